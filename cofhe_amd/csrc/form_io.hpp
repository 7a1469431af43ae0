// form_io.hpp -- record <-> register moves (global memory), see layout.hpp
#pragma once
#include "layout.hpp"
#include "qf.hpp"
namespace cofhe {

CF_DEV void qf_load(Ctx &c, QForm &f, const uint32_t *rec) {
    CF_UNROLL for (int j = 0; j < CH; j++) {
        f.a.v[0][j] = rec[REC_A + c.gl * CH + j];
        f.bm.v[0][j] = rec[REC_B + c.gl * CH + j];
        f.c.v[0][j] = rec[REC_C + c.gl * CH + j];
        f.c.v[1][j] = rec[REC_C + PLIMBS + c.gl * CH + j];
    }
    f.bneg = (int)rec[REC_SIGN];
}

CF_DEV void qf_store(Ctx &c, const QForm &f, uint32_t *rec) {
    CF_UNROLL for (int j = 0; j < CH; j++) {
        rec[REC_A + c.gl * CH + j] = f.a.v[0][j];
        rec[REC_B + c.gl * CH + j] = f.bm.v[0][j];
        rec[REC_C + c.gl * CH + j] = f.c.v[0][j];
        rec[REC_C + PLIMBS + c.gl * CH + j] = f.c.v[1][j];
    }
    if (c.gl == 0) rec[REC_SIGN] = (uint32_t)f.bneg;
    if (c.gl > 0) rec[REC_SIGN + c.gl] = 0u;
}

}  // namespace cofhe
