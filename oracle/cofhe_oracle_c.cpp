// C entry points of the CPU oracle (ctypes-loadable).  TEST INFRASTRUCTURE ONLY -- see the
// header of cofhe_oracle.hpp.  All tensors travel in the reference's binary formats
// (ciphertext tensors: cpu_cryptosystem.inl:320-392; plaintext tensors: :229-267).
#include "cofhe_oracle.hpp"

#include <chrono>
#include <cstdlib>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace cofhe_oracle;

namespace {
Z z_from_le(const uint8_t *p, size_t n, int negative) {
    Z v;
    mpz_import(v.get_mpz_t(), n, -1, 1, -1, 0, p);
    if (negative) v = -v;
    return v;
}
int emit(const std::string &s, uint8_t **out, size_t *outlen) {
    *out = (uint8_t *)malloc(s.size() ? s.size() : 1);
    if (!*out) return -2;
    memcpy(*out, s.data(), s.size());
    *outlen = s.size();
    return 0;
}
thread_local std::string g_err;
}  // namespace

extern "C" {

const char *oracle_last_error() { return g_err.c_str(); }
void oracle_free(uint8_t *p) { free(p); }
int oracle_max_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void oracle_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

// delta is passed as the little-endian magnitude of |Delta| (Delta itself is negative)
// mode 0: nucomp (reference path), mode 1: Gauss composition + reduction (cross-check)
int oracle_add_ciphertext_tensors(const uint8_t *absdelta, size_t dlen, const uint8_t *t1, size_t l1,
                                  const uint8_t *t2, size_t l2, int mode, uint8_t **out, size_t *outlen) {
    try {
        ClassGroup G(z_from_le(absdelta, dlen, 1));
        std::vector<uint32_t> s1, s2;
        CtVec a = deserialize_ciphertext_tensor(std::string((const char *)t1, l1), s1);
        CtVec b = deserialize_ciphertext_tensor(std::string((const char *)t2, l2), s2);
        if (s1 != s2) { free_cts(a); free_cts(b); throw std::invalid_argument("Tensor shapes must be equal"); }
        CtVec r;
        if (mode == 0) {
            r = add_ciphertext_tensors(G, a, b);
        } else {
            r.resize(a.size());
            for (size_t i = 0; i < a.size(); i++) {
                r[i] = new CipherText();
                G.compose_gauss(r[i]->c1, a[i]->c1, b[i]->c1);
                G.compose_gauss(r[i]->c2, a[i]->c2, b[i]->c2);
            }
        }
        int rc = emit(serialize_ciphertext_tensor(s1, r), out, outlen);
        free_cts(a); free_cts(b); free_cts(r);
        return rc;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

// 1-D x 1-D branch (tensor_ops.inl:280-340); s in plaintext-tensor format
int oracle_scal_ciphertext_tensors_1d(const uint8_t *absdelta, size_t dlen, const uint8_t *s, size_t ls,
                                      const uint8_t *t, size_t lt, uint8_t **out, size_t *outlen) {
    try {
        ClassGroup G(z_from_le(absdelta, dlen, 1));
        std::vector<uint32_t> ss, st;
        std::vector<Z> sv = deserialize_plaintext_tensor(std::string((const char *)s, ls), ss);
        CtVec a = deserialize_ciphertext_tensor(std::string((const char *)t, lt), st);
        if (sv.size() != a.size()) { free_cts(a); throw std::invalid_argument("Vector sizes must be equal"); }
        CtVec r = scal_ciphertext_tensors_1d(G, sv, a);
        int rc = emit(serialize_ciphertext_tensor(st, r), out, outlen);
        free_cts(a); free_cts(r);
        return rc;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

// 2-D branch (tensor_ops.inl:342-461); zero = 1-element ciphertext tensor
int oracle_scal_ciphertext_tensors_2d(const uint8_t *absdelta, size_t dlen, const uint8_t *s, size_t ls,
                                      const uint8_t *t, size_t lt, const uint8_t *zero, size_t lz,
                                      uint8_t **out, size_t *outlen) {
    try {
        ClassGroup G(z_from_le(absdelta, dlen, 1));
        std::vector<uint32_t> ss, st, sz;
        std::vector<Z> sv = deserialize_plaintext_tensor(std::string((const char *)s, ls), ss);
        CtVec a = deserialize_ciphertext_tensor(std::string((const char *)t, lt), st);
        CtVec z = deserialize_ciphertext_tensor(std::string((const char *)zero, lz), sz);
        if (ss.size() != 2 || st.size() != 2 || z.size() != 1 || st[1] != ss[0]) {
            free_cts(a); free_cts(z);
            throw std::invalid_argument("Tensors must be 0D, 1D or 2D for now");
        }
        size_t n = st[0], m = st[1], p = ss[1];
        CtVec r = scal_ciphertext_tensors_2d(G, sv, a, *z[0], n, m, p);
        int rc = emit(serialize_ciphertext_tensor({(uint32_t)n, (uint32_t)p}, r), out, outlen);
        free_cts(a); free_cts(z); free_cts(r);
        return rc;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

// batched shared-table exponentiation of ONE base (qfi.inl:1-135): base = c1 of the single
// ciphertext in `t`; returns a count-element tensor whose c1 = base^{s_i} and c2 = c2^{s_i}
int oracle_qfi_nupow(const uint8_t *absdelta, size_t dlen, const uint8_t *s, size_t ls, const uint8_t *t,
                     size_t lt, uint8_t **out, size_t *outlen) {
    try {
        ClassGroup G(z_from_le(absdelta, dlen, 1));
        std::vector<uint32_t> ss, st;
        std::vector<Z> sv = deserialize_plaintext_tensor(std::string((const char *)s, ls), ss);
        CtVec a = deserialize_ciphertext_tensor(std::string((const char *)t, lt), st);
        if (a.size() != 1) { free_cts(a); throw std::invalid_argument("one base expected"); }
        std::vector<const Z *> sp;
        for (auto &z : sv) sp.push_back(&z);
        std::vector<QFI> r1, r2;
        qfi_nupow(r1, G, a[0]->c1, sp.data(), sp.size());
        qfi_nupow(r2, G, a[0]->c2, sp.data(), sp.size());
        CtVec r(sv.size());
        for (size_t i = 0; i < r.size(); i++) r[i] = new CipherText{r1[i], r2[i]};
        int rc = emit(serialize_ciphertext_tensor({(uint32_t)r.size()}, r), out, outlen);
        free_cts(a); free_cts(r);
        return rc;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

// returns 1 when every form in the tensor is reduced and has discriminant Delta
int oracle_check_tensor(const uint8_t *absdelta, size_t dlen, const uint8_t *t, size_t lt) {
    try {
        Z delta = z_from_le(absdelta, dlen, 1);
        std::vector<uint32_t> st;
        CtVec a = deserialize_ciphertext_tensor(std::string((const char *)t, lt), st);
        int ok = 1;
        for (auto *ct : a)
            for (const QFI *f : {&ct->c1, &ct->c2}) {
                Z d = f->b * f->b - 4 * f->a * f->c;
                if (d != delta || !ClassGroup::is_reduced(*f)) ok = 0;
            }
        free_cts(a);
        return ok;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

// plaintext encoding; writes the decimal string of make_plaintext(x) into buf
int oracle_make_plaintext(float x, uint32_t k, char *buf, size_t cap) {
    std::string s = make_plaintext(x, k).get_str(10);
    if (s.size() + 1 > cap) return -1;
    memcpy(buf, s.c_str(), s.size() + 1);
    return 0;
}
float oracle_get_float_from_plaintext(const char *dec, uint32_t k) { return get_float_from_plaintext(Z(dec), k); }

// ---- CPU baseline timing (bench.py cpu_baseline leg) -------------------------------------
// One timed run of `chain` chained tensor adds res = add(res, t2) starting from add(t1, t2),
// with the per-iteration frees of benchmarks/local.cpp:99-117.  Returns seconds, <0 on error.
double oracle_time_matadd_chain(const uint8_t *absdelta, size_t dlen, const uint8_t *t1, size_t l1,
                                const uint8_t *t2, size_t l2, int chain, uint8_t **out, size_t *outlen) {
    try {
        ClassGroup G(z_from_le(absdelta, dlen, 1));
        std::vector<uint32_t> s1, s2;
        CtVec a = deserialize_ciphertext_tensor(std::string((const char *)t1, l1), s1);
        CtVec b = deserialize_ciphertext_tensor(std::string((const char *)t2, l2), s2);
        auto t0 = std::chrono::steady_clock::now();
        CtVec res = add_ciphertext_tensors(G, a, b);
        for (int i = 1; i < chain; i++) {
            CtVec rc = add_ciphertext_tensors(G, res, b);
            free_cts(res);
            res = std::move(rc);
        }
        auto t1e = std::chrono::steady_clock::now();
        if (out) emit(serialize_ciphertext_tensor(s1, res), out, outlen);
        free_cts(a); free_cts(b); free_cts(res);
        return std::chrono::duration<double>(t1e - t0).count();
    } catch (const std::exception &e) { g_err = e.what(); return -1.0; }
}

double oracle_time_scal_2d(const uint8_t *absdelta, size_t dlen, const uint8_t *s, size_t ls, const uint8_t *t,
                           size_t lt, const uint8_t *zero, size_t lz) {
    try {
        ClassGroup G(z_from_le(absdelta, dlen, 1));
        std::vector<uint32_t> ss, st, sz;
        std::vector<Z> sv = deserialize_plaintext_tensor(std::string((const char *)s, ls), ss);
        CtVec a = deserialize_ciphertext_tensor(std::string((const char *)t, lt), st);
        CtVec z = deserialize_ciphertext_tensor(std::string((const char *)zero, lz), sz);
        size_t n = st[0], m = st[1], p = ss[1];
        auto t0 = std::chrono::steady_clock::now();
        CtVec r = scal_ciphertext_tensors_2d(G, sv, a, *z[0], n, m, p);
        auto t1e = std::chrono::steady_clock::now();
        free_cts(a); free_cts(z); free_cts(r);
        return std::chrono::duration<double>(t1e - t0).count();
    } catch (const std::exception &e) { g_err = e.what(); return -1.0; }
}

}  // extern "C"
