// local_bench.cpp -- counterpart of the reference's local harness (benchmarks/local.cpp:65-215,
// benchmarks/benchmark.hpp) on the MI355X engine: same construction of inputs
// (make_plaintext(i+1), encrypt_tensor), same 1+49 chained operations with the per-iteration
// frees, first/last/average/median/total milliseconds per tag.  Additionally reports
// ciphertext-ops/s, the same chain with tensors kept resident in HBM, and writes the serialised
// bytes of the final tensor so a parity checker can compare them.
//
//   ./local_bench encrypt_decrypt [n m]          (reference default 64 64)
//   ./local_bench ciphertext_matadd [n m]        (reference default 64 64)
//   ./local_bench scal_matmul [n m p [chain]]    (reference default 8 64 64, 50 chained products)
//   ./local_bench threshold [n m t parties]      (threshold decryption, default 16 16 2 3)
//   ./local_bench ciphertext_matmul [n m p [t parties]]   (Beaver-triplet ct x ct product, default 4 4 4)
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <ctime>
#include <fstream>
#include <iostream>
#include <memory>
#include <numeric>
#include <thread>
#include <utility>

#include "hip_cryptosystem.hpp"
#include "smpc_local.hpp"

using namespace CoFHE;
using Clock = std::chrono::steady_clock;

// Same bookkeeping as the reference's Benchmark (benchmarks/benchmark.hpp:5-146): start / end stamp of every run, the
// printed summary, and save() -- the results file ./benchmark_results_<tag><date>.txt with the summary block and one
// "Start: .. End: .." line per run (benchmark.hpp:96-116).  The harness calls save() when LOCAL_BENCH_SAVE is set.
struct Benchmark {
    using TP = std::chrono::time_point<std::chrono::high_resolution_clock>;
    std::string tag;
    std::vector<std::pair<TP, TP>> timestamps;
    std::vector<double> ms;
    explicit Benchmark(std::string t) : tag(std::move(t)) {}
    template <typename F>
    void run(F &&f, int times) {
        for (int i = 0; i < times; i++) {
            const TP t0 = std::chrono::high_resolution_clock::now();
            f();
            const TP t1 = std::chrono::high_resolution_clock::now();
            timestamps.emplace_back(t0, t1);
            ms.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
        }
    }
    double total() const { return std::accumulate(ms.begin(), ms.end(), 0.0); }
    double median() const {
        std::vector<double> s = ms;
        std::sort(s.begin(), s.end());
        return s[s.size() / 2];
    }
    void summary(std::ostream &o) const {
        o << "======================" << std::endl;
        o << "Benchmark summary " + tag << std::endl;
        o << "Number of runs: " << ms.size() << std::endl;
        o << "First run time: " << ms.front() << "ms" << std::endl;
        o << "Last run time: " << ms.back() << "ms" << std::endl;
        o << "Average time: " << total() / ms.size() << "ms" << std::endl;
        o << "Median time: " << median() << "ms" << std::endl;
        o << "Total time: " << total() << "ms" << std::endl;
        o << "======================" << std::endl;
    }
    void print_summary() const {
        if (ms.empty()) return;
        std::cout << "Benchmark: " << tag << "\n  runs " << ms.size() << " first " << ms.front() << " ms, last " << ms.back()
                  << " ms, average " << total() / ms.size() << " ms, median " << median() << " ms, total " << total()
                  << " ms" << std::endl;
        if (getenv("LOCAL_BENCH_SAVE")) save();
    }
    std::string save() const {
        if (ms.empty()) return "";
        time_t now = time(nullptr);
        struct tm tstruct = *localtime(&now);
        char buf[80];
        strftime(buf, sizeof(buf), "%Y-%m-%d.%X", &tstruct);
        std::string name = tag;
        for (char &ch : name)
            if (ch == ' ' || ch == '/' || ch == ':' || ch == '(' || ch == ')' || ch == '<' || ch == '>' || ch == '*') ch = '_';
        const std::string filename = "./benchmark_results_" + name + buf + ".txt";
        std::ofstream file(filename, std::ios::app);
        summary(file);
        for (const auto &se : timestamps)
            file << "Start: " << se.first.time_since_epoch().count() << " End: " << se.second.time_since_epoch().count() << std::endl;
        return filename;
    }
};

template <typename T>
static void free_all(Tensor<T *> t) {
    t.flatten();
    for (size_t i = 0; i < t.num_elements(); i++) delete t.at(i);
}

static void bench_matadd(size_t n, size_t m) {
    auto cs = make_cryptosystem(128, 128, Device::GPU);
    using CS = decltype(cs);
    auto sk = cs.keygen();
    auto pk = cs.keygen(sk);
    Tensor<CS::PlainText *> pt1(n, m, nullptr), pt2(n, m, nullptr);
    pt1.flatten(); pt2.flatten();
    for (size_t i = 0; i < n * m; i++) {
        pt1.at(i) = new CS::PlainText(cs.make_plaintext(i + 1));
        pt2.at(i) = new CS::PlainText(cs.make_plaintext(i + 1));
    }
    pt1.reshape({n, m}); pt2.reshape({n, m});
    auto ct1 = cs.encrypt_tensor(pk, pt1);
    auto ct2 = cs.encrypt_tensor(pk, pt2);
    Benchmark b("ciphertext_matadd (API: Tensor<CipherText*> in and out each op)");
    std::string final_bytes;
    double api_chain_ms = 0, api_ser_ms = 0;
    b.run([&]() {
        auto t0 = Clock::now();
        auto res = cs.add_ciphertext_tensors(pk, ct1, ct2);
        for (int i = 0; i < 49; ++i) {
            auto res_c = cs.add_ciphertext_tensors(pk, res, ct2);
            free_all(res);
            res = res_c;
        }
        cs.synchronize();
        auto t1 = Clock::now();
        final_bytes = cs.serialize_ciphertext_tensor(res);
        free_all(res);
        api_chain_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        api_ser_ms = std::chrono::duration<double, std::milli>(Clock::now() - t1).count();
    }, 1);
    std::cout << "  of which 50 API calls " << api_chain_ms << " ms, serialising the result " << api_ser_ms << " ms" << std::endl;
    b.print_summary();
    std::cout << "  " << 50.0 * n * m / (b.ms[0] * 1e-3) << " ciphertext-ops/s (host marshalling included)" << std::endl;
    Benchmark r("ciphertext_matadd (tensors resident in HBM)");
    std::string resident_bytes;
    r.run([&]() {
        auto d1 = cs.upload(ct1);
        auto d2 = cs.upload(ct2);
        auto t0 = Clock::now();
        auto res = cs.add_ciphertext_tensors(d1, d2);
        for (int i = 0; i < 49; ++i) res = cs.add_ciphertext_tensors(res, d2);
        cs.synchronize();
        double ms = std::chrono::duration<double, std::milli>(Clock::now() - t0).count();
        std::cout << "  resident chain of 50: " << ms << " ms, " << 50.0 * n * m / (ms * 1e-3) << " ciphertext-ops/s" << std::endl;
        auto back = cs.download(res);
        resident_bytes = cs.serialize_ciphertext_tensor(back);
        free_all(back);
    }, 1);
    r.print_summary();
    std::cout << "  API chain and resident chain agree: " << (final_bytes == resident_bytes ? "yes" : "NO") << std::endl;
    std::ofstream("local_bench_matadd_ct1.bin", std::ios::binary) << cs.serialize_ciphertext_tensor(ct1);
    std::ofstream("local_bench_matadd_ct2.bin", std::ios::binary) << cs.serialize_ciphertext_tensor(ct2);
    std::ofstream("local_bench_matadd_out.bin", std::ios::binary) << final_bytes;
    {
        Mpz ad = cs.discriminant();
        ad.neg();
        std::ofstream("local_bench_absdelta.txt") << ad.str() << "\n";
    }
    free_all(pt1); free_all(pt2); free_all(ct1); free_all(ct2);
    std::cout << "n: " << n << " m: " << m << std::endl;
}

static void bench_scal_matmul(size_t n, size_t m, size_t p, int chain) {
    auto cs = make_cryptosystem(128, 128, Device::GPU);
    using CS = decltype(cs);
    auto sk = cs.keygen();
    auto pk = cs.keygen(sk);
    Tensor<CS::PlainText *> pt1(n, m, nullptr), pt2(m, p, nullptr);
    pt1.flatten(); pt2.flatten();
    for (size_t i = 0; i < n * m; i++) pt1.at(i) = new CS::PlainText(cs.make_plaintext(i + 1));
    for (size_t i = 0; i < m * p; i++) pt2.at(i) = new CS::PlainText(cs.make_plaintext(i + 1));
    pt1.reshape({n, m}); pt2.reshape({m, p});
    auto ct1 = cs.encrypt_tensor(pk, pt1);
    // the reference draws a fresh Enc(0) inside every product (tensor_ops.inl:352); a fixed one makes the run
    // reproducible for the parity checker
    auto zero = cs.encrypt(pk, cs.make_plaintext(0));
    if (m != p) chain = 1;                                   // the chain re-feeds the n x p result as the n x m operand
    Benchmark b("scal_matmul (1 + " + std::to_string(chain - 1) + " chained products, benchmarks/local.cpp:177-197)");
    std::string final_bytes;
    b.run([&]() {
        auto res = cs.scal_ciphertext_tensors(pk, pt2, ct1, &zero);
        for (int i = 0; i < chain - 1; ++i) {
            auto res_c = cs.scal_ciphertext_tensors(pk, pt2, res, &zero);
            free_all(res);
            res = res_c;
        }
        final_bytes = cs.serialize_ciphertext_tensor(res);
        free_all(res);
    }, 1);
    b.print_summary();
    std::cout << "  " << (double)chain * n * p / (b.ms[0] * 1e-3) << " output ciphertexts/s" << std::endl;
    std::ofstream("local_bench_scal_s.bin", std::ios::binary) << cs.serialize_plaintext_tensor(pt2);
    std::ofstream("local_bench_scal_cts.bin", std::ios::binary) << cs.serialize_ciphertext_tensor(ct1);
    {
        Tensor<CS::CipherText *> zt(1, &zero);
        std::ofstream("local_bench_scal_zero.bin", std::ios::binary) << cs.serialize_ciphertext_tensor(zt);
    }
    std::ofstream("local_bench_scal_out.bin", std::ios::binary) << final_bytes;
    {
        Mpz ad = cs.discriminant();
        ad.neg();
        std::ofstream("local_bench_absdelta.txt") << ad.str() << "\n";
    }
    free_all(pt1); free_all(pt2); free_all(ct1);
    std::cout << "n: " << n << " m: " << m << " p: " << p << " chain: " << chain << std::endl;
}

// counterpart of benchmark_encrypt_decrypt (benchmarks/local.cpp:22-63), with the check the
// reference omits: every plaintext must come back
static void bench_encrypt_decrypt(size_t n, size_t m) {
    auto cs = make_cryptosystem(128, 128, Device::GPU);
    using CS = decltype(cs);
    auto sk = cs.keygen();
    auto pk = cs.keygen(sk);
    Tensor<CS::PlainText *> pts(n, m, nullptr);
    pts.flatten();
    for (size_t i = 0; i < n * m; i++) pts.at(i) = new CS::PlainText(cs.make_plaintext(i + 1));
    pts.reshape({n, m});
    Benchmark b("encrypt_decrypt");
    bool ok = true;
    double enc_first_ms = 0, enc_again_ms = 0, dec_ms = 0;
    b.run([&]() {
        // whole encrypt_tensor calls, h^r and pk^r included: the first builds the fixed-base tables of h and pk
        // (one chain of ~1000 squarings each), later calls find them in the context
        auto t0 = Clock::now();
        auto ct0 = cs.encrypt_tensor(pk, pts);
        cs.synchronize();
        auto t1 = Clock::now();
        auto ct = cs.encrypt_tensor(pk, pts);
        cs.synchronize();
        auto t2 = Clock::now();
        free_all(ct0);
        auto res = cs.decrypt_tensor(sk, ct);
        auto t3 = Clock::now();
        enc_first_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        enc_again_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
        dec_ms = std::chrono::duration<double, std::milli>(t3 - t2).count();
        ct.flatten(); res.flatten();
        for (size_t i = 0; i < ct.num_elements(); i++) {
            if (cs.get_float_from_plaintext(*res.at(i)) != (float)(i + 1)) ok = false;
            delete res.at(i);
            delete ct.at(i);
        }
    }, 1);
    // homomorphic identities through the whole stack: Dec(Enc a + Enc b) = a + b, Dec(3 * Enc a) = 3a
    {
        auto a = cs.encrypt(pk, cs.make_plaintext(230)), bb = cs.encrypt(pk, cs.make_plaintext(20));
        auto sum = cs.add_ciphertexts(pk, a, bb);
        auto tri = cs.scal_ciphertext(pk, cs.make_plaintext(3), a);
        auto neg = cs.scal_ciphertext(pk, cs.make_plaintext(-1), a);
        if (cs.get_float_from_plaintext(cs.decrypt(sk, sum)) != 250.0f) ok = false;
        if (cs.get_float_from_plaintext(cs.decrypt(sk, tri)) != 690.0f) ok = false;
        if (cs.get_float_from_plaintext(cs.decrypt(sk, neg)) != -230.0f) ok = false;
    }
    // the 0-D tensor branches (tensor_ops.inl:199-202, 275-278) call the scalar forms: randomised like the reference's ...
    {
        auto a = cs.encrypt(pk, cs.make_plaintext(7)), bb = cs.encrypt(pk, cs.make_plaintext(5));
        auto three = cs.make_plaintext(3);
        Tensor<CS::CipherText *> ta(&a), tb(&bb);
        Tensor<CS::PlainText *> ts(&three);
        auto sum = cs.add_ciphertext_tensors(pk, ta, tb);
        auto tri = cs.scal_ciphertext_tensors(pk, ts, ta);
        if (!sum.is_zero_degree() || !tri.is_zero_degree()) ok = false;
        if (cs.get_float_from_plaintext(cs.decrypt(sk, *sum.get_value())) != 12.0f) ok = false;
        if (cs.get_float_from_plaintext(cs.decrypt(sk, *tri.get_value())) != 21.0f) ok = false;
        // a second call gives a different ciphertext of the same value (fresh r)
        auto sum2 = cs.add_ciphertext_tensors(pk, ta, tb);
        if (sum2.get_value()->c1() == sum.get_value()->c1()) ok = false;
        delete sum.get_value(); delete sum2.get_value(); delete tri.get_value();
        // ... and deterministic when asked, so that a checker can compare bytes
        cs.set_rerandomize(false);
        auto dsum = cs.add_ciphertext_tensors(pk, ta, tb);
        auto dtri = cs.scal_ciphertext_tensors(pk, ts, ta);
        auto one = [&](const CS::CipherText &c) { CS::CipherText cc = c; Tensor<CS::CipherText *> t(1, &cc); return cs.serialize_ciphertext_tensor(t); };
        std::ofstream("local_bench_scalar_a.bin", std::ios::binary) << one(a);
        std::ofstream("local_bench_scalar_b.bin", std::ios::binary) << one(bb);
        std::ofstream("local_bench_scalar_sum.bin", std::ios::binary) << one(*dsum.get_value());
        std::ofstream("local_bench_scalar_tri.bin", std::ios::binary) << one(*dtri.get_value());
        delete dsum.get_value(); delete dtri.get_value();
        cs.set_rerandomize(true);
        Mpz ad = cs.discriminant();
        ad.neg();
        std::ofstream("local_bench_absdelta.txt") << ad.str() << "\n";
    }
    b.print_summary();
    free_all(pts);
    std::cout << "  encrypt_tensor (h^r, pk^r included): first call " << enc_first_ms << " ms (builds the tables of h and pk), next call "
              << enc_again_ms << " ms = " << n * m / (enc_again_ms * 1e-3) << " ciphertexts/s; decrypt_tensor " << dec_ms << " ms" << std::endl;
    std::cout << "  roundtrip and homomorphic checks: " << (ok ? "ok" : "FAILED") << std::endl;
    std::cout << "n: " << n << " m: " << m << std::endl;
    if (!ok) throw std::runtime_error("decryption mismatch");
}

// threshold decryption end to end (the reference has no local benchmark for it; the calls are the
// ones PartialDecryptionRequestHandler / SMPCClient make, partial_decryption_request_handler.hpp:140,
// smpc_client.hpp:137): share sk t-out-of-n, every party of the first threshold set runs
// part_decrypt_tensor, the combiner runs combine_part_decryption_results_tensor.
static void bench_threshold(size_t n, size_t m, size_t t, size_t parties) {
    auto cs = make_cryptosystem(128, 128, Device::GPU);
    using CS = decltype(cs);
    auto sk = cs.keygen();
    auto pk = cs.keygen(sk);
    auto shares = cs.keygen(sk, t, parties);
    // host check of the sharing: for the first threshold set {0..t-1}, s_0 - s_1 - ... - s_(t-1) = sk
    {
        Mpz acc = shares[0][0];
        for (size_t j = 1; j < t; j++) mpz_sub(acc.get(), acc.get(), shares[j][0].get());
        if (!(acc == sk)) throw std::runtime_error("shares do not reconstruct the secret key");
    }
    Tensor<CS::PlainText *> pts(n, m, nullptr);
    pts.flatten();
    for (size_t i = 0; i < n * m; i++) pts.at(i) = new CS::PlainText(cs.make_plaintext(i + 1));
    pts.reshape({n, m});
    auto ct = cs.encrypt_tensor(pk, pts);
    Benchmark bp("part_decrypt_tensor"), bc("combine_part_decryption_results_tensor");
    Vector<Tensor<CS::PartDecryptionResult *>> pdrs;
    bp.run([&]() { pdrs.push_back(cs.part_decrypt_tensor(shares[pdrs.size()][0], ct)); }, (int)t);
    // wire format round trip of the first party's result
    {
        auto bytes = cs.serialize_part_decryption_result_tensor(pdrs[0]);
        auto back = cs.deserialize_part_decryption_result_tensor(bytes);
        if (cs.serialize_part_decryption_result_tensor(back) != bytes) throw std::runtime_error("pdr format round trip");
        free_all(back);
    }
    bool ok = true;
    bc.run([&]() {
        auto res = cs.combine_part_decryption_results_tensor(ct, pdrs);
        res.flatten();
        for (size_t i = 0; i < res.num_elements(); i++) {
            if (cs.get_float_from_plaintext(*res.at(i)) != (float)(i + 1)) ok = false;
            delete res.at(i);
        }
    }, 1);
    bp.print_summary();
    bc.print_summary();
    for (auto &p : pdrs) free_all(p);
    free_all(ct);
    free_all(pts);
    std::cout << "  threshold " << t << " of " << parties << ": " << (ok ? "ok" : "FAILED") << std::endl;
    std::cout << "n: " << n << " m: " << m << std::endl;
    if (!ok) throw std::runtime_error("threshold decryption mismatch");
}

// ciphertext x ciphertext matrix product through the Beaver-triplet protocol with an in-process
// client (smpc_local.hpp; reference: SMPCCipherTextMultiplier::multiply_ciphertext_tensors,
// include/smpc/ciphertext_multiplications.hpp:40-112).  threshold = 0: the client decrypts with the
// secret key; otherwise by t-of-n threshold decryption.
static void bench_ciphertext_matmul(size_t n, size_t m, size_t p, size_t t, size_t parties) {
    auto cs = make_cryptosystem(128, 128, Device::GPU);
    using CS = decltype(cs);
    auto sk = cs.keygen();
    std::unique_ptr<LocalSMPCClient<CS>> client(t ? new LocalSMPCClient<CS>(cs, sk, t, parties) : new LocalSMPCClient<CS>(cs, sk));
    LocalCipherTextMultiplier<CS> mul(*client);
    const auto &pk = client->network_public_key();
    Tensor<CS::PlainText *> pa(n, m, nullptr), pb(m, p, nullptr);
    pa.flatten(); pb.flatten();
    for (size_t i = 0; i < n * m; i++) pa.at(i) = new CS::PlainText(cs.make_plaintext((float)(i % 7) - 3.0f));
    for (size_t i = 0; i < m * p; i++) pb.at(i) = new CS::PlainText(cs.make_plaintext((float)(i % 5) + 1.0f));
    pa.reshape({n, m}); pb.reshape({m, p});
    auto ca = cs.encrypt_tensor(pk, pa), cb = cs.encrypt_tensor(pk, pb);
    Benchmark b("ciphertext_matmul (Beaver triplets, in-process client)");
    bool ok = true;
    b.run([&]() {
        auto res = mul.multiply_ciphertext_tensors(ca, cb);
        auto dec = cs.decrypt_tensor(sk, res);
        res.flatten(); dec.flatten();
        for (size_t i = 0; i < n; i++)
            for (size_t k = 0; k < p; k++) {
                float want = 0;
                for (size_t j = 0; j < m; j++) want += ((float)((i * m + j) % 7) - 3.0f) * ((float)((j * p + k) % 5) + 1.0f);
                if (cs.get_float_from_plaintext(*dec.at(i * p + k)) != want) ok = false;
            }
        free_all(res);
        free_all(dec);
    }, 1);
    b.print_summary();
    std::cout << "  " << n * m * p << " element products, " << client->decrypted_elements() << " decryptions"
              << (t ? " (threshold " + std::to_string(t) + " of " + std::to_string(parties) + ")" : std::string(" (secret key)"))
              << ": " << (ok ? "ok" : "FAILED") << std::endl;
    std::cout << "n: " << n << " m: " << m << " p: " << p << std::endl;
    free_all(ca); free_all(cb); free_all(pa); free_all(pb);
    if (!ok) throw std::runtime_error("ciphertext matmul mismatch");
}

// text formats of single values and the binary plaintext-tensor format (cpu_cryptosystem.inl:124-318): written to
// files for the parity checker, and read back
static void bench_formats() {
    auto cs = make_cryptosystem(128, 128, Device::GPU);
    using CS = decltype(cs);
    auto sk = cs.keygen();
    auto pk = cs.keygen(sk);
    auto ct = cs.encrypt(pk, cs.make_plaintext(42));
    bool ok = true;
    const std::string txt = cs.serialize_ciphertext(ct);
    auto back = cs.deserialize_ciphertext(txt);
    if (!(back.c1() == ct.c1()) || !(back.c2() == ct.c2())) ok = false;
    if (!(cs.deserialize_public_key(cs.serialize_public_key(pk)) == pk)) ok = false;
    if (!(cs.deserialize_secret_key(cs.serialize_secret_key(sk)) == sk)) ok = false;
    auto pd = cs.part_decrypt(sk, ct);
    if (!(cs.deserialize_part_decryption_result(cs.serialize_part_decryption_result(pd)) == pd)) ok = false;
    Tensor<CS::PlainText *> pts(2, 3, nullptr);
    pts.flatten();
    const float vals[6] = {0.0f, 1.0f, -1.0f, 255.0f, -65536.0f, 123456.0f};
    for (int i = 0; i < 6; i++) pts.at(i) = new CS::PlainText(cs.make_plaintext(vals[i]));
    pts.reshape({2, 3});
    const std::string bin = cs.serialize_plaintext_tensor(pts);
    auto pb = cs.deserialize_plaintext_tensor(bin);
    if (pb.shape() != pts.shape()) ok = false;
    for (int i = 0; i < 6; i++)
        if (!(*pb[i] == *pts[i])) ok = false;
    if (cs.serialize_plaintext_tensor(pb) != bin) ok = false;
    if (cs.serialize_plaintext(*pts[3]) != "255" || !(cs.deserialize_plaintext("255") == *pts[3])) ok = false;
    {
        CS::CipherText cc = ct;
        Tensor<CS::CipherText *> t(1, &cc);
        std::ofstream("local_bench_fmt_ct.bin", std::ios::binary) << cs.serialize_ciphertext_tensor(t);
    }
    std::ofstream("local_bench_fmt_ct.txt") << txt;
    std::ofstream("local_bench_fmt_pt.bin", std::ios::binary) << bin;
    {
        std::ofstream f("local_bench_fmt_pt.txt");
        for (int i = 0; i < 6; i++) f << cs.serialize_plaintext(*pts[i]) << "\n";
    }
    free_all(pb);
    // ---- the rest of the reference's surface on this path (cpu_cryptosystem.hpp:103-104, 127, 139)
    {
        Mpz bound;
        mpz_setbit(bound.get(), cs.message_bits());
        for (int i = 0; i < 8; i++) {
            auto rp = cs.generate_random_plaintext();
            if (rp.sgn() < 0 || mpz_cmp(rp.get(), bound.get()) >= 0) ok = false;
            auto tr = cs.generate_random_beavers_triplet();
            if (tr.size() != 3 || mpz_cmp_ui(tr[0].get(), 10) >= 0 || mpz_cmp_ui(tr[1].get(), 10) >= 0) ok = false;
            if (!(cs.multiply_plaintexts(tr[0], tr[1]) == tr[2])) ok = false;
        }
        auto again = CS::deserialize(cs.serialize());
        if (again.message_bits() != cs.message_bits() || again.serialize() != cs.serialize()) ok = false;
        // in-place accumulation through the class-group handles, as the node layer writes it
        // (include/smpc/ciphertext_multiplications.hpp:85-98), against add_ciphertexts without re-randomisation
        auto x = cs.encrypt(pk, cs.make_plaintext(5)), y = cs.encrypt(pk, cs.make_plaintext(9));
        cs.set_rerandomize(false);
        auto want = cs.add_ciphertexts(pk, x, y);
        cs.set_rerandomize(true);
        CS::CipherText *res = new CS::CipherText(x);
        auto cl_g = cs.get_hsm2k().Cl_G();
        auto cl_delta = cs.get_hsm2k().Cl_Delta();
        cl_g.nucomp(res->c1(), res->c1(), y.c1());
        cl_delta.nucomp(res->c2(), res->c2(), y.c2());
        if (!(std::as_const(*res).c1() == want.c1()) || !(std::as_const(*res).c2() == want.c2())) ok = false;
        if (cs.get_float_from_plaintext(cs.decrypt(sk, *res)) != 14.0f) ok = false;
        // nucompinv undoes nucomp; nudupl and nupow agree; the principal form is neutral
        QFI back2, sq, p2, neutral;
        cl_g.nucompinv(back2, std::as_const(*res).c1(), y.c1());
        if (!(back2 == x.c1())) ok = false;
        cl_g.nudupl(sq, x.c1());
        cl_g.nupow(p2, x.c1(), Mpz(2ul));
        if (!(sq == p2)) ok = false;
        cl_g.nucomp(neutral, x.c2(), cl_g.one());
        if (!(neutral == x.c2())) ok = false;
        delete res;
        // negate_plaintext_tensor keeps the shape
        Tensor<CS::PlainText *> np = cs.negate_plaintext_tensor(pts);
        if (np.shape() != pts.shape() || cs.get_float_from_plaintext(*np[3]) != -255.0f) ok = false;
        free_all(np);
    }
    free_all(pts);
    // the enum factories
    auto cs2 = make_cryptosystem(SecurityLevel::MEDIUM, 128, Device::GPU);
    auto cs3 = make_cryptosystem(SecurityLevel::MEDIUM, Precision::FP32, 2, Device::GPU);
    if (cs2.message_bits() != 128 || cs3.message_bits() != 128) ok = false;
    std::cout << "  formats: " << (ok ? "ok" : "FAILED") << std::endl;
    if (!ok) throw std::runtime_error("format round trip mismatch");
}

// one HIPCryptoSystem shared by two host threads (the reference's compute server calls one instance from 8 threads,
// include/node/server.hpp:16,185-197): a matrix product (workspace + tables of the context) next to decryptions
// (same workspace, cached table of f), results compared with the single-threaded ones
static void bench_threads(int rounds) {
    auto cs = make_cryptosystem(128, 128, Device::GPU);
    using CS = decltype(cs);
    auto sk = cs.keygen();
    auto pk = cs.keygen(sk);
    const size_t n = 3, m = 4, p = 4;
    Tensor<CS::PlainText *> pt1(n, m, nullptr), pt2(m, p, nullptr);
    pt1.flatten(); pt2.flatten();
    for (size_t i = 0; i < n * m; i++) pt1.at(i) = new CS::PlainText(cs.make_plaintext(i + 1));
    for (size_t i = 0; i < m * p; i++) pt2.at(i) = new CS::PlainText(cs.make_plaintext((float)(i % 5) - 2.0f));
    pt1.reshape({n, m}); pt2.reshape({m, p});
    auto ct1 = cs.encrypt_tensor(pk, pt1);
    auto zero = cs.encrypt(pk, cs.make_plaintext(0));
    auto ref = cs.scal_ciphertext_tensors(pk, pt2, ct1, &zero);
    const std::string ref_bytes = cs.serialize_ciphertext_tensor(ref);
    std::atomic<bool> ok{true};
    std::thread ta([&]() {
        for (int r = 0; r < rounds; r++) {
            // alternate shapes so that the workspace is re-sized while the other thread uses the context
            auto res = cs.scal_ciphertext_tensors(pk, pt2, ct1, &zero);
            if (cs.serialize_ciphertext_tensor(res) != ref_bytes) ok = false;
            free_all(res);
        }
    });
    std::thread tb([&]() {
        for (int r = 0; r < rounds; r++) {
            auto dec = cs.decrypt_tensor(sk, ct1);
            dec.flatten();
            for (size_t i = 0; i < n * m; i++) {
                if (cs.get_float_from_plaintext(*dec.at(i)) != (float)(i + 1)) ok = false;
                delete dec.at(i);
            }
        }
    });
    ta.join();
    tb.join();
    // A shared RESULT tensor read by both threads the way the reference reads it -- through the tensor's non-const element
    // pointers, cts.at(i)->c1() (cpu_cryptosystem_tensor_ops.inl:175; the compute server shares tensors between its 8
    // threads): the non-const accessors must not touch the element's block reference (they used to reset it: a race with
    // the other reader and with copies).  Lazy elements, first touched concurrently; every value compared with a
    // single-threaded const read of a second, identical result.
    {
        auto shared = cs.scal_ciphertext_tensors(pk, pt2, ct1, &zero);
        auto check = cs.scal_ciphertext_tensors(pk, pt2, ct1, &zero);
        shared.flatten(); check.flatten();
        const size_t E = n * p;
        std::vector<std::string> want(E);
        for (size_t i = 0; i < E; i++) {
            const CS::CipherText &c = *check.at(i);
            want[i] = c.c1().a().str() + " " + c.c2().b().str();
        }
        auto reader = [&](int start) {
            for (int r = 0; r < rounds; r++)
                for (size_t k = 0; k < E; k++) {
                    const size_t i = (k + start) % E;
                    CS::CipherText *e = shared.at(i);                 // non-const, as a Tensor<CipherText *> hands it out
                    CS::CipherText copy(*e);                          // copies race with the other thread's first read
                    if (e->c1().a().str() + " " + e->c2().b().str() != want[i]) ok = false;
                    if (copy.c1().a().str() + " " + copy.c2().b().str() != want[i]) ok = false;
                }
        };
        std::thread r1(reader, 0), r2(reader, (int)(E / 2));
        r1.join();
        r2.join();
        // a touched element no longer offers its device copy; the values are still what the block holds
        for (size_t i = 0; i < E; i++)
            if (shared.at(i)->block()) ok = false;
        free_all(shared); free_all(check);
    }
    free_all(ref); free_all(ct1); free_all(pt1); free_all(pt2);
    std::cout << "  two threads on one cryptosystem, " << rounds << " rounds each: " << (ok ? "ok" : "FAILED") << std::endl;
    if (!ok) throw std::runtime_error("concurrent use gave a different result");
}

// make_plaintext / get_float_from_plaintext of the product on floats given by their bit patterns (one 8-digit hex word
// per line of `in`): "<decimal plaintext> <bit pattern of the float that comes back>" per line of `out`
static void plaintexts_mode(const char *in, const char *out) {
    auto cs = make_cryptosystem(128, 128, Device::GPU);
    std::ifstream fi(in);
    std::ofstream fo(out);
    std::string tok;
    while (fi >> tok) {
        const uint32_t bits = (uint32_t)std::stoul(tok, nullptr, 16);
        float x;
        memcpy(&x, &bits, 4);
        auto pt = cs.make_plaintext(x);
        const float back = cs.get_float_from_plaintext(pt);
        uint32_t bb;
        memcpy(&bb, &back, 4);
        char buf[16];
        snprintf(buf, sizeof buf, "%08x", bb);
        fo << pt.str() << " " << buf << "\n";
    }
}

int main(int argc, char **argv) {
    if (argc < 2) {
        std::cerr << "Usage: " << argv[0] << " <encrypt_decrypt|ciphertext_matadd|scal_matmul|threshold|ciphertext_matmul> [sizes]" << std::endl;
        return 1;
    }
    std::string mode = argv[1];
    try {
        if (mode == "ciphertext_matadd") {
            size_t n = argc > 2 ? std::stoul(argv[2]) : 64, m = argc > 3 ? std::stoul(argv[3]) : 64;
            bench_matadd(n, m);
        } else if (mode == "encrypt_decrypt") {
            size_t n = argc > 2 ? std::stoul(argv[2]) : 64, m = argc > 3 ? std::stoul(argv[3]) : 64;
            bench_encrypt_decrypt(n, m);
        } else if (mode == "scal_matmul") {
            size_t n = argc > 2 ? std::stoul(argv[2]) : 8, m = argc > 3 ? std::stoul(argv[3]) : 64,
                   p = argc > 4 ? std::stoul(argv[4]) : 64;
            const int chain = argc > 5 ? std::stoi(argv[5]) : 50;
            bench_scal_matmul(n, m, p, chain);
        } else if (mode == "ciphertext_matmul") {
            size_t n = argc > 2 ? std::stoul(argv[2]) : 4, m = argc > 3 ? std::stoul(argv[3]) : 4,
                   p = argc > 4 ? std::stoul(argv[4]) : 4, t = argc > 5 ? std::stoul(argv[5]) : 0,
                   parties = argc > 6 ? std::stoul(argv[6]) : 3;
            bench_ciphertext_matmul(n, m, p, t, parties);
        } else if (mode == "plaintexts") {
            if (argc < 4) throw std::invalid_argument("plaintexts <in> <out>");
            plaintexts_mode(argv[2], argv[3]);
        } else if (mode == "formats") {
            bench_formats();
        } else if (mode == "threads") {
            bench_threads(argc > 2 ? std::stoi(argv[2]) : 4);
        } else if (mode == "threshold") {
            size_t n = argc > 2 ? std::stoul(argv[2]) : 16, m = argc > 3 ? std::stoul(argv[3]) : 16,
                   t = argc > 4 ? std::stoul(argv[4]) : 2, parties = argc > 5 ? std::stoul(argv[5]) : 3;
            bench_threshold(n, m, t, parties);
        } else {
            std::cerr << "Invalid benchmark type" << std::endl;
            return 1;
        }
    } catch (const std::exception &e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 2;
    }
    return 0;
}
