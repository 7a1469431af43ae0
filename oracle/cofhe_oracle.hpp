// oracle/cofhe_oracle.hpp -- CPU restatement (C++17 + GMP) of CoFHE's local
// ciphertext-tensor path.  TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load this; nothing under cofhe_amd/ or include/ may.
//
// PARITY UNPINNED.  The arithmetic of the reference lives in thirdparty/bicycl, an empty and
// unpinned submodule (/root/reference/.gitmodules:4-6, CMakeLists.txt:38), and the reference
// holds no tests / golden vectors.  This file restates the published
// algorithms the reference's call sites name; it is pinned only by (i) the uniqueness of the
// reduced form of a class, (ii) oracle/pyref.py (independent pure-Python big-int model) on the
// committed fixtures in tests/golden/, (iii) algebraic identities in tests/.
//
// Reference lines followed:
//   add_ciphertext_tensors      include/x86_64/cpu_cryptosystem_tensor_ops.inl:197-267
//   scal_ciphertext_tensors     ...tensor_ops.inl:269-462 (0-D guard, 1-D branch, 2-D branch)
//   qfi_nupow (shared wNAF-7)   include/x86_64/qfi.inl:1-135
//   byte format F / P           include/x86_64/cpu_cryptosystem.inl:320-508 / :229-318
//   make_plaintext / map_back   include/x86_64/cpu_cryptosystem.inl:49-87, hpp:150-161
//   nucomp / nudupl / reduction Cohen, CCANT Alg. 5.4.2, 5.4.7, 5.4.8, 5.4.9 (BICYCL absent)
#pragma once
#include <gmpxx.h>

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace cofhe_oracle {

using Z = mpz_class;

struct QFI {
    Z a, b, c;
    bool operator==(const QFI &o) const { return a == o.a && b == o.b && c == o.c; }
};

struct CipherText {
    QFI c1, c2;
};

inline size_t nbits(const Z &x) { return mpz_sgn(x.get_mpz_t()) == 0 ? 0 : mpz_sizeinbase(x.get_mpz_t(), 2); }

// ---- Lehmer partial Euclid (what BICYCL's Mpz::partial_euclid provides to nucomp) --------
// On entry R0 > R1 >= 0, (C0, C1) cofactors with R_i == C_i * r (mod modulus).  Runs the
// Euclidean remainder sequence until bits(R1) <= stop_bits (or R1 == 0).  Knuth's Algorithm L
// on the leading 63 bits, in-place updates on caller-provided scratch (no allocation in the
// hot loop).
struct EuclidScratch {
    Z q, t, u, v;
};
inline void partial_euclid(Z &R0, Z &R1, Z &C0, Z &C1, size_t stop_bits, EuclidScratch &S) {
    Z &q = S.q, &t = S.t, &u = S.u, &v = S.v;
    while (mpz_sgn(R1.get_mpz_t()) != 0 && nbits(R1) > stop_bits) {
        size_t n = nbits(R0);
        bool did = false;
        if (n > 64 && nbits(R1) > stop_bits + 64) {
            size_t sh = n - 62;
            mpz_tdiv_q_2exp(t.get_mpz_t(), R0.get_mpz_t(), sh);
            int64_t x = (int64_t)mpz_get_ui(t.get_mpz_t());
            mpz_tdiv_q_2exp(t.get_mpz_t(), R1.get_mpz_t(), sh);
            int64_t y = (int64_t)mpz_get_ui(t.get_mpz_t());
            // cofactors: (x_i) = A x + B y with alternating signs; keep |entries| < 2^31
            int64_t A = 1, B = 0, C = 0, D = 1;
            while (true) {
                // Knuth 4.5.2 Algorithm L: the quotient is certain when both extremes agree
                if (y + C == 0 || y + D == 0) break;
                int64_t q1 = (x + A) / (y + C), q2 = (x + B) / (y + D);
                if (q1 != q2) break;
                int64_t nC = A - q1 * C, nD = B - q1 * D;
                if (nC > (1LL << 30) || nC < -(1LL << 30) || nD > (1LL << 30) || nD < -(1LL << 30)) break;
                int64_t r = x - q1 * y;
                A = C; B = D; C = nC; D = nD;
                x = y; y = r;
            }
            if (B != 0) {
                // (R0, R1) <- (A R0 + B R1, C R0 + D R1), same for the cofactors
                mpz_mul_si(u.get_mpz_t(), R0.get_mpz_t(), A);
                mpz_mul_si(t.get_mpz_t(), R1.get_mpz_t(), B);
                mpz_add(u.get_mpz_t(), u.get_mpz_t(), t.get_mpz_t());
                mpz_mul_si(v.get_mpz_t(), R0.get_mpz_t(), C);
                mpz_mul_si(t.get_mpz_t(), R1.get_mpz_t(), D);
                mpz_add(v.get_mpz_t(), v.get_mpz_t(), t.get_mpz_t());
                mpz_swap(R0.get_mpz_t(), u.get_mpz_t());
                mpz_swap(R1.get_mpz_t(), v.get_mpz_t());
                mpz_mul_si(u.get_mpz_t(), C0.get_mpz_t(), A);
                mpz_mul_si(t.get_mpz_t(), C1.get_mpz_t(), B);
                mpz_add(u.get_mpz_t(), u.get_mpz_t(), t.get_mpz_t());
                mpz_mul_si(v.get_mpz_t(), C0.get_mpz_t(), C);
                mpz_mul_si(t.get_mpz_t(), C1.get_mpz_t(), D);
                mpz_add(v.get_mpz_t(), v.get_mpz_t(), t.get_mpz_t());
                mpz_swap(C0.get_mpz_t(), u.get_mpz_t());
                mpz_swap(C1.get_mpz_t(), v.get_mpz_t());
                did = true;
            }
        }
        if (!did) {
            mpz_tdiv_qr(q.get_mpz_t(), t.get_mpz_t(), R0.get_mpz_t(), R1.get_mpz_t());
            mpz_swap(R0.get_mpz_t(), R1.get_mpz_t());
            mpz_swap(R1.get_mpz_t(), t.get_mpz_t());
            mpz_submul(C0.get_mpz_t(), q.get_mpz_t(), C1.get_mpz_t());
            mpz_swap(C0.get_mpz_t(), C1.get_mpz_t());
        }
    }
}
inline void partial_euclid(Z &R0, Z &R1, Z &C0, Z &C1, size_t stop_bits) {
    EuclidScratch S;
    partial_euclid(R0, R1, C0, C1, stop_bits, S);
}

class ClassGroup {
  public:
    explicit ClassGroup(const Z &delta) : delta_(delta) {
        if (mpz_sgn(delta.get_mpz_t()) >= 0) throw std::invalid_argument("discriminant must be negative");
        Z ad = -delta;
        // default_nucomp_bound(): floor(|Delta/4|^(1/4)) (call sites tensor_ops.inl:394-395)
        Z q = ad / 4;
        mpz_root(bound_.get_mpz_t(), q.get_mpz_t(), 4);
        stop_bits_ = nbits(bound_);
    }
    const Z &discriminant() const { return delta_; }
    const Z &default_nucomp_bound() const { return bound_; }

    QFI one() const {
        QFI r;
        r.a = 1;
        r.b = mpz_odd_p(delta_.get_mpz_t()) ? 1 : 0;
        r.c = (r.b - delta_) / 4;
        return r;
    }

    // Cohen 5.4.2 with -a < b <= a; b >= 0 when a == c
    static void normalize(QFI &f) {
        Z q, two_a = 2 * f.a, t = f.a - f.b;
        mpz_fdiv_q(q.get_mpz_t(), t.get_mpz_t(), two_a.get_mpz_t());
        q = -q;
        if (mpz_sgn(q.get_mpz_t()) != 0) {
            f.c = f.a * q * q - f.b * q + f.c;
            f.b = f.b - two_a * q;
        }
    }
    static void reduce(QFI &f) {
        while (true) {
            normalize(f);
            int cmp = mpz_cmp(f.a.get_mpz_t(), f.c.get_mpz_t());
            if (cmp > 0) {
                std::swap(f.a, f.c);
                f.b = -f.b;
                continue;
            }
            if (cmp == 0 && mpz_sgn(f.b.get_mpz_t()) < 0) f.b = -f.b;
            return;
        }
    }
    static bool is_reduced(const QFI &f) {
        Z na = -f.a;
        if (!(na < f.b && f.b <= f.a && f.a <= f.c)) return false;
        return !(f.a == f.c && mpz_sgn(f.b.get_mpz_t()) < 0);
    }
    static void inverse(QFI &r, const QFI &f) {
        r = f;
        r.b = -r.b;
        reduce(r);
    }

    // Cohen 5.4.7 + reduction: general composition (any gcd structure)
    void compose_gauss(QFI &r, const QFI &f1_, const QFI &f2_) const {
        const QFI *p1 = &f1_, *p2 = &f2_;
        if (p1->a > p2->a) std::swap(p1, p2);
        const Z &a1 = p1->a, &b1 = p1->b, &a2 = p2->a, &b2 = p2->b, &c2 = p2->c;
        Z s = (b1 + b2) / 2, n = b2 - s;
        Z y1, y2, x2, d, d1, u, v;
        if (mpz_divisible_p(a2.get_mpz_t(), a1.get_mpz_t())) {
            y1 = 0; d = a1;
        } else {
            mpz_gcdext(d.get_mpz_t(), u.get_mpz_t(), v.get_mpz_t(), a2.get_mpz_t(), a1.get_mpz_t());
            y1 = u;
        }
        if (mpz_divisible_p(s.get_mpz_t(), d.get_mpz_t())) {
            y2 = -1; x2 = 0; d1 = d;
        } else {
            mpz_gcdext(d1.get_mpz_t(), u.get_mpz_t(), v.get_mpz_t(), s.get_mpz_t(), d.get_mpz_t());
            x2 = u; y2 = -v;
        }
        Z v1 = a1 / d1, v2 = a2 / d1;
        Z rr = y1 * y2 * n - x2 * c2;
        mpz_mod(rr.get_mpz_t(), rr.get_mpz_t(), v1.get_mpz_t());
        QFI o;
        o.b = b2 + 2 * v2 * rr;
        o.a = v1 * v2;
        Z num = c2 * d1 + rr * (b2 + v2 * rr);
        mpz_divexact(o.c.get_mpz_t(), num.get_mpz_t(), v1.get_mpz_t());
        reduce(o);
        r = std::move(o);
    }

    // NUCOMP (Cohen 5.4.9 / Jacobson-van der Poorten) for every gcd structure:
    //   d = gcd(a1,a2) = y1 a2 (mod a1), d1 = gcd(s,d) = x2 s - y2 d, v1 = a1/d1, v2 = a2/d1,
    //   r = (y1 y2 (-m) - x2 c2) mod v1, partial Euclid on (v1, r), then for a pair (R, C):
    //   M1 = (v2 R - m C)/v1, M2 = (s R + c2 d1 C)/v1, a' = R1 M1 + C1 M2,
    //   b' = -sign(det) 2 (R0 M1 + C0 M2) - b1, c' = (b'^2 - Delta)/(4 a'); reduce.
    // negf2: compose with f2^-1 (qfi.inl:111,127 pass `neg`).
    void nucomp(QFI &r, const QFI &f1_, const QFI &f2_, bool negf2 = false) const {
        QFI f2n;
        const QFI *p1 = &f1_, *p2 = &f2_;
        if (negf2) { f2n = f2_; f2n.b = -f2n.b; p2 = &f2n; }
        if (p1->a < p2->a) std::swap(p1, p2);
        const Z &a1 = p1->a, &b1 = p1->b, &a2 = p2->a, &b2 = p2->b, &c2 = p2->c;
        static thread_local EuclidScratch S;
        static thread_local Z d, y1, m, s, v1, v2, c2d, rr, R0, R1, C0, C1, M1, M2, t, t2;
        mpz_gcdext(d.get_mpz_t(), y1.get_mpz_t(), nullptr, a2.get_mpz_t(), a1.get_mpz_t());
        mpz_sub(m.get_mpz_t(), b1.get_mpz_t(), b2.get_mpz_t());
        mpz_tdiv_q_2exp(m.get_mpz_t(), m.get_mpz_t(), 1);            // exact
        mpz_add(s.get_mpz_t(), b1.get_mpz_t(), b2.get_mpz_t());
        mpz_tdiv_q_2exp(s.get_mpz_t(), s.get_mpz_t(), 1);
        if (mpz_cmp_ui(d.get_mpz_t(), 1) == 0) {
            v1 = a1; v2 = a2; c2d = c2;
            mpz_mul(rr.get_mpz_t(), y1.get_mpz_t(), m.get_mpz_t());
            mpz_mod(rr.get_mpz_t(), rr.get_mpz_t(), v1.get_mpz_t());
        } else {
            Z d1, x2, y2;
            if (mpz_divisible_p(s.get_mpz_t(), d.get_mpz_t())) {
                d1 = d; x2 = 0; y2 = -1;
            } else {
                Z vv;
                mpz_gcdext(d1.get_mpz_t(), x2.get_mpz_t(), vv.get_mpz_t(), s.get_mpz_t(), d.get_mpz_t());
                y2 = -vv;
            }
            mpz_divexact(v1.get_mpz_t(), a1.get_mpz_t(), d1.get_mpz_t());
            mpz_divexact(v2.get_mpz_t(), a2.get_mpz_t(), d1.get_mpz_t());
            c2d = c2 * d1;
            rr = -(y1 * y2 * m) - x2 * c2;
            mpz_mod(rr.get_mpz_t(), rr.get_mpz_t(), v1.get_mpz_t());
        }
        // adaptive bound keeps a', c' near sqrt|Delta| for every size of v1, v2
        long stop = ((long)nbits(v1) - (long)nbits(v2) + (long)(nbits(delta_) + 1) / 2) / 2;
        R0 = v1; R1 = rr; C0 = 0; C1 = 1;
        partial_euclid(R0, R1, C0, C1, stop < 0 ? 0 : (size_t)stop, S);
        int sg = mpz_sgn(C1.get_mpz_t());
        // M1, M2 from (R1, C1)
        mpz_mul(M1.get_mpz_t(), v2.get_mpz_t(), R1.get_mpz_t());
        mpz_submul(M1.get_mpz_t(), m.get_mpz_t(), C1.get_mpz_t());
        mpz_divexact(M1.get_mpz_t(), M1.get_mpz_t(), v1.get_mpz_t());
        mpz_mul(M2.get_mpz_t(), s.get_mpz_t(), R1.get_mpz_t());
        mpz_addmul(M2.get_mpz_t(), c2d.get_mpz_t(), C1.get_mpz_t());
        mpz_divexact(M2.get_mpz_t(), M2.get_mpz_t(), v1.get_mpz_t());
        QFI o;
        mpz_mul(o.a.get_mpz_t(), R1.get_mpz_t(), M1.get_mpz_t());
        mpz_addmul(o.a.get_mpz_t(), C1.get_mpz_t(), M2.get_mpz_t());
        mpz_mul(t.get_mpz_t(), R0.get_mpz_t(), M1.get_mpz_t());
        mpz_addmul(t.get_mpz_t(), C0.get_mpz_t(), M2.get_mpz_t());
        mpz_mul_2exp(t.get_mpz_t(), t.get_mpz_t(), 1);
        if (sg > 0) mpz_neg(t.get_mpz_t(), t.get_mpz_t());
        mpz_sub(o.b.get_mpz_t(), t.get_mpz_t(), b1.get_mpz_t());
        mpz_mul(t2.get_mpz_t(), o.b.get_mpz_t(), o.b.get_mpz_t());
        mpz_sub(t2.get_mpz_t(), t2.get_mpz_t(), delta_.get_mpz_t());
        mpz_mul_2exp(t.get_mpz_t(), o.a.get_mpz_t(), 2);
        mpz_divexact(o.c.get_mpz_t(), t2.get_mpz_t(), t.get_mpz_t());
        reduce(o);
        r = std::move(o);
    }

    // NUDUPL (Cohen 5.4.8): the same formulas with f1 == f2 (d = a, so the general branch runs)
    void nudupl(QFI &r, const QFI &f) const { nucomp(r, f, f); }

    // ClassGroup::nupow: f^n reduced (plain left-to-right binary; n may be <= 0)
    void nupow(QFI &r, const QFI &f, const Z &n) const {
        int sgn = mpz_sgn(n.get_mpz_t());
        if (sgn == 0) { r = one(); return; }
        Z e = abs(n);
        QFI acc = f;
        for (long i = (long)nbits(e) - 2; i >= 0; --i) {
            nudupl(acc, acc);
            if (mpz_tstbit(e.get_mpz_t(), i)) nucomp(acc, acc, f);
        }
        if (sgn < 0) inverse(acc, acc);
        r = std::move(acc);
    }

  private:
    Z delta_, bound_;
    size_t stop_bits_;
};

// qfi_nupow: one base, `count` exponents, shared wNAF-7 table and doubling cache, restating
// include/x86_64/qfi.inl:1-135 (own code; same algorithm so the CPU baseline does the same
// work).  Results are the reduced f^{n_i}; n_i == 0 yields the identity (the reference's code
// would read bit -1 there; the mathematically correct power is returned instead).
inline void qfi_nupow(std::vector<QFI> &out, const ClassGroup &G, const QFI &f, const Z *const *n, size_t count) {
    out.assign(count, QFI());
    if (count == 0) return;
    const unsigned w = 7;
    const unsigned long pow2w = 1UL << w, u = 1UL << (w - 2);
    QFI ff;
    G.nudupl(ff, f);
    std::vector<QFI> tab(u);
    tab[0] = f;
    for (unsigned long i = 1; i < u; i++) G.nucomp(tab[i], tab[i - 1], ff);
    std::unordered_map<std::string, QFI> cache;   // keyed by the exact degree (no size_t wrap)
    auto key = [](const Z &d) { return d.get_str(16); };
    auto get_doubled = [&](QFI &r, Z &deg, size_t k) {
        Z fin = deg << k;
        auto kd = key(deg);
        if (!cache.count(kd)) cache[kd] = r;
        auto it = cache.find(key(fin));
        if (it != cache.end()) {
            deg = fin;
            r = it->second;
            return;
        }
        for (size_t i = 0; i < k; i++) {
            G.nudupl(r, r);
            deg <<= 1;
            cache[key(deg)] = r;
        }
    };
    auto extract = [](const Z &x, long j, unsigned len) -> unsigned long {
        // bits j, j-1, ..., j-len+1 of x as an integer (missing low bits read as 0)
        unsigned long v = 0;
        for (unsigned t = 0; t < len; t++) {
            long pos = j - (long)t;
            v = (v << 1) | (pos >= 0 ? (unsigned long)mpz_tstbit(x.get_mpz_t(), pos) : 0UL);
        }
        return v;
    };
    for (size_t idx = 0; idx < count; idx++) {
        Z e = abs(*n[idx]);
        QFI &r = out[idx];
        if (mpz_sgn(e.get_mpz_t()) == 0) { r = G.one(); continue; }
        Z deg;
        long j = (long)nbits(e) - 1;
        unsigned long c;
        {
            unsigned long m = extract(e, j, w);
            c = m & 1;
            unsigned long t = m + (m & 1);
            size_t val2 = __builtin_ctzl(t);
            size_t tau = val2 < w ? val2 : w - 1;
            t >>= tau;
            r = (t == 2) ? ff : tab[t >> 1];
            deg = (t == 2) ? 2UL : ((t >> 1) * 2 + 1);
            size_t b = ((size_t)j) < w - 1 ? tau + 1 + j - w : tau;
            get_doubled(r, deg, b);
            j -= w;
        }
        while (j >= 0) {
            unsigned long m = extract(e, j, w);
            unsigned long dj = (m >> (w - 1)) & 1, djmwp1 = m & 1;
            if (c == dj) {
                get_doubled(r, deg, 1);
                j -= 1;
            } else {
                bool neg = c != 0;
                unsigned long t = m + djmwp1;
                t = c ? (pow2w - t) : t;
                c = djmwp1;
                size_t val2 = t > 0 ? (size_t)__builtin_ctzl(t) : w - 1;
                size_t tau = val2 < w ? val2 : w - 1;
                t >>= tau;
                get_doubled(r, deg, w - tau);
                G.nucomp(r, r, (t == 2) ? ff : tab[t >> 1], neg);
                unsigned long dd = (t == 2) ? 2 : ((t >> 1) * 2 + 1);
                if (neg) deg -= dd; else deg += dd;
                size_t b = ((size_t)j) < w - 1 ? tau + 1 + j - w : tau;
                get_doubled(r, deg, b);
                j -= w;
            }
        }
        if (c) G.nucomp(r, r, tab[0], true);
        if (mpz_sgn(n[idx]->get_mpz_t()) < 0) ClassGroup::inverse(r, r);
    }
}

// ---- tensor ops: loop structure (and per-element heap allocation) of the reference ---------
using CtVec = std::vector<CipherText *>;

inline void free_cts(CtVec &v) {
    for (auto *p : v) delete p;
    v.clear();
}

// tensor_ops.inl:242-264
inline CtVec add_ciphertext_tensors(const ClassGroup &G, const CtVec &ct1, const CtVec &ct2) {
    if (ct1.size() != ct2.size()) throw std::invalid_argument("Tensor shapes must be equal");
    CtVec res(ct1.size(), nullptr);
    long E = (long)ct1.size();
#pragma omp parallel for schedule(static)
    for (long i = 0; i < E; i++) {
        QFI c1, c2;
        G.nucomp(c1, ct1[i]->c1, ct2[i]->c1);
        G.nucomp(c2, ct1[i]->c2, ct2[i]->c2);
        res[i] = new CipherText{std::move(c1), std::move(c2)};
    }
    return res;
}

// tensor_ops.inl:316-338 (1-D x 1-D)
inline CtVec scal_ciphertext_tensors_1d(const ClassGroup &G, const std::vector<Z> &s, const CtVec &cts) {
    if (s.size() != cts.size()) throw std::invalid_argument("Vector sizes must be equal");
    CtVec res(cts.size(), nullptr);
    long E = (long)cts.size();
#pragma omp parallel for schedule(static)
    for (long i = 0; i < E; i++) {
        QFI c1, c2;
        G.nupow(c1, cts[i]->c1, s[i]);
        G.nupow(c2, cts[i]->c2, s[i]);
        res[i] = new CipherText{std::move(c1), std::move(c2)};
    }
    return res;
}

// tensor_ops.inl:342-461 (2-D): cts n x m, s m x p, `zero` passed in (the reference draws it
// from its RandGen at :352; parity inputs therefore carry it explicitly)
inline CtVec scal_ciphertext_tensors_2d(const ClassGroup &G, const std::vector<Z> &s, const CtVec &cts,
                                        const CipherText &zero, size_t n, size_t m, size_t p) {
    CtVec res(n * p, nullptr);
    for (size_t i = 0; i < n * p; i++) res[i] = new CipherText(zero);
    std::vector<const Z *> sp(m * p);
    for (size_t i = 0; i < m * p; i++) sp[i] = &s[i];
    std::vector<std::vector<QFI>> p1(n * m), p2(n * m);
    long NM = (long)(n * m);
#pragma omp parallel for schedule(static)
    for (long ij = 0; ij < NM; ij++) {
        size_t j = (size_t)ij % m;
        qfi_nupow(p1[ij], G, cts[ij]->c1, sp.data() + j * p, p);
        qfi_nupow(p2[ij], G, cts[ij]->c2, sp.data() + j * p, p);
    }
    long NP = (long)(n * p);
#pragma omp parallel for schedule(static)
    for (long ik = 0; ik < NP; ik++) {
        size_t i = (size_t)ik / p, k = (size_t)ik % p;
        for (size_t j = 0; j < m; j++) {
            G.nucomp(res[ik]->c1, res[ik]->c1, p1[i * m + j][k]);
            G.nucomp(res[ik]->c2, res[ik]->c2, p2[i * m + j][k]);
        }
    }
    return res;
}

// ---- byte formats (cpu_cryptosystem.inl:320-508 ciphertext tensors, :229-318 plaintexts) ---
inline size_t slot_width(const Z &x) { return mpz_sizeinbase(x.get_mpz_t(), 2) / 8 + 1; }

inline std::string serialize_ints(const std::vector<uint32_t> &shape, const std::vector<const Z *> &vals) {
    size_t cnt = vals.size();
    std::vector<uint64_t> offs(cnt);
    uint64_t last = 0;
    for (size_t i = 0; i < cnt; i++) {
        offs[i] = last | (mpz_sgn(vals[i]->get_mpz_t()) != 1 ? (1ULL << 63) : 0ULL);
        last += slot_width(*vals[i]);
    }
    std::string data(4 + 4 * shape.size() + 8 * cnt + last, '\0');
    char *p = data.data();
    uint32_t ndim = (uint32_t)shape.size();
    memcpy(p, &ndim, 4); p += 4;
    for (uint32_t d : shape) { memcpy(p, &d, 4); p += 4; }
    memcpy(p, offs.data(), 8 * cnt); p += 8 * cnt;
    for (size_t i = 0; i < cnt; i++)
        mpz_export(p + (offs[i] & ~(1ULL << 63)), nullptr, -1, 1, -1, 0, vals[i]->get_mpz_t());
    return data;
}

inline void deserialize_ints(const std::string &data, size_t per_elem, std::vector<uint32_t> &shape, std::vector<Z> &vals) {
    if (data.size() < 4) throw std::invalid_argument("short buffer");
    const char *p = data.data();
    uint32_t ndim;
    memcpy(&ndim, p, 4); p += 4;
    shape.resize(ndim);
    uint64_t ne = 1;
    for (uint32_t i = 0; i < ndim; i++) { memcpy(&shape[i], p, 4); p += 4; ne *= shape[i]; }
    size_t cnt = ne * per_elem;
    size_t hdr = 4 + 4 * ndim + 8 * cnt;
    if (data.size() < hdr) throw std::invalid_argument("short buffer");
    std::vector<uint64_t> offs(cnt);
    memcpy(offs.data(), p, 8 * cnt); p += 8 * cnt;
    size_t body = data.size() - hdr;
    vals.assign(cnt, Z());
    const uint64_t M = ~(1ULL << 63);
    for (size_t i = 0; i < cnt; i++) {
        uint64_t st = offs[i] & M, en = (i + 1 < cnt) ? (offs[i + 1] & M) : body;
        if (en < st || en > body) throw std::invalid_argument("bad offsets");
        mpz_import(vals[i].get_mpz_t(), en - st, -1, 1, -1, 0, p + st);
        if (offs[i] >> 63) vals[i] = -vals[i];
    }
}

inline std::string serialize_ciphertext_tensor(const std::vector<uint32_t> &shape, const CtVec &cts) {
    std::vector<const Z *> v;
    v.reserve(cts.size() * 6);
    for (auto *ct : cts) {
        v.push_back(&ct->c1.a); v.push_back(&ct->c1.b); v.push_back(&ct->c1.c);
        v.push_back(&ct->c2.a); v.push_back(&ct->c2.b); v.push_back(&ct->c2.c);
    }
    return serialize_ints(shape, v);
}

inline CtVec deserialize_ciphertext_tensor(const std::string &data, std::vector<uint32_t> &shape) {
    std::vector<Z> vals;
    deserialize_ints(data, 6, shape, vals);
    CtVec out(vals.size() / 6);
    for (size_t i = 0; i < out.size(); i++) {
        out[i] = new CipherText{QFI{vals[6 * i], vals[6 * i + 1], vals[6 * i + 2]},
                                QFI{vals[6 * i + 3], vals[6 * i + 4], vals[6 * i + 5]}};
    }
    return out;
}

inline std::string serialize_plaintext_tensor(const std::vector<uint32_t> &shape, const std::vector<Z> &pts) {
    std::vector<const Z *> v;
    for (auto &z : pts) v.push_back(&z);
    return serialize_ints(shape, v);
}

inline std::vector<Z> deserialize_plaintext_tensor(const std::string &data, std::vector<uint32_t> &shape) {
    std::vector<Z> vals;
    deserialize_ints(data, 1, shape, vals);
    return vals;
}

// ---- plaintext encoding (cpu_cryptosystem.inl:49-87, hpp:150-161): same GMP calls ----------
inline Z make_plaintext(float x, uint32_t k) {
    mpf_t sx, M, sf;
    mpf_init(sx); mpf_init(M); mpf_init(sf);
    mpf_set_d(sf, 2); mpf_set_d(M, 2);
    mpf_pow_ui(sf, sf, 0);
    mpf_pow_ui(M, M, k);
    mpf_set_d(sx, x);
    mpf_mul(sx, sx, sf);
    if (x < 0) mpf_add(sx, sx, M);
    Z r;
    mpz_set_f(r.get_mpz_t(), sx);
    mpf_clear(sx); mpf_clear(M); mpf_clear(sf);
    return r;
}

inline float get_float_from_plaintext(const Z &z, uint32_t k) {
    mpf_t num, M, Mh;
    mpf_init(num); mpf_init(M); mpf_init(Mh);
    mpf_set_d(M, 2);
    mpf_pow_ui(M, M, k);
    mpf_div_ui(Mh, M, 2);
    mpf_set_z(num, z.get_mpz_t());
    if (mpf_cmp(num, Mh) >= 0) mpf_sub(num, num, M);
    float r = (float)mpf_get_d(num);
    mpf_clear(num); mpf_clear(M); mpf_clear(Mh);
    return r;
}

}  // namespace cofhe_oracle
