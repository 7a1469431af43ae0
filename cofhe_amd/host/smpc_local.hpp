// smpc_local.hpp -- the LOCAL part of CoFHE's ciphertext x ciphertext multiplication (Beaver
// triplets) on the MI355X engine.
//
// Reference: SMPCCipherTextMultiplier<CryptoSystem> (include/smpc/ciphertext_multiplications.hpp:8-175)
// runs every tensor operation of the protocol on the caller's CryptoSystem and reaches the
// network only through its SMPCClient for two things: get_beavers_triplets(n) and
// decrypt_tensor(ct) (threshold decryption by the CoFHE nodes).  Here that client is a template
// parameter; LocalSMPCClient below answers both calls in-process (triplets from fresh
// randomness, decryption either with the secret key or by t-of-n threshold decryption:
// part_decrypt_tensor per party + combine_part_decryption_results_tensor), so the whole
// multiplication is the sequence of GPU kernels the networked system would run, minus sockets.
// The protocol per element (ciphertext_multiplications.hpp:115-160), with (a, b, c = ab) a triplet:
//   e = Dec(x - a), d = Dec(y - b);  x*y = e*b + d*a + c + e*d
// i.e. 2 negations, 5 additions, 2 scalings, 1 encryption and 2 decryptions per product.
#pragma once
#include "hip_cryptosystem.hpp"

namespace CoFHE {

// What the multiplier needs from "the network".  Mirrors the members of SMPCClient it calls
// (include/smpc/smpc_client.hpp: crypto_system(), network_public_key(), get_beavers_triplets,
// decrypt, decrypt_tensor).
template <typename CryptoSystem>
class LocalSMPCClient {
  public:
    using SecretKey = typename CryptoSystem::SecretKey;
    using SecretKeyShare = typename CryptoSystem::SecretKeyShare;
    using PublicKey = typename CryptoSystem::PublicKey;
    using PlainText = typename CryptoSystem::PlainText;
    using CipherText = typename CryptoSystem::CipherText;
    using PartDecryptionResult = typename CryptoSystem::PartDecryptionResult;

    // single key holder
    LocalSMPCClient(CryptoSystem &cs, const SecretKey &sk) : cs_(cs), sk_(sk), pk_(cs.keygen(sk)) {}
    // t-of-n: decryption goes through the threshold path with the first threshold set {0..t-1}
    LocalSMPCClient(CryptoSystem &cs, const SecretKey &sk, size_t threshold, size_t parties)
        : cs_(cs), sk_(sk), pk_(cs.keygen(sk)), threshold_(threshold) {
        auto shares = cs.keygen(sk, threshold, parties);
        for (size_t j = 0; j < threshold; j++) shares_.push_back(shares[j][0]);
    }

    CryptoSystem &crypto_system() { return cs_; }
    const PublicKey &network_public_key() const { return pk_; }

    // n x 3 tensor of (Enc a, Enc b, Enc ab), a and b uniform in the message space Z/2^k (the
    // reference's generator, include/smpc/beavers_triplet_generation.hpp, is a network protocol
    // between the nodes)
    Tensor<CipherText *> get_beavers_triplets(size_t n) {
        Tensor<PlainText *> pa(n, nullptr), pb(n, nullptr);
        for (size_t i = 0; i < n; i++) {
            pa[i] = new PlainText(cs_.random_plaintext(cs_.message_bits()));
            pb[i] = new PlainText(cs_.random_plaintext(cs_.message_bits()));
        }
        auto pc = cs_.multiply_plaintext_tensors(pa, pb);
        auto ea = cs_.encrypt_tensor(pk_, pa), eb = cs_.encrypt_tensor(pk_, pb), ec = cs_.encrypt_tensor(pk_, pc);
        Tensor<CipherText *> t(n, 3, nullptr);
        for (size_t i = 0; i < n; i++) {
            t.at(i, 0) = ea[i];
            t.at(i, 1) = eb[i];
            t.at(i, 2) = ec[i];
            delete pa[i];
            delete pb[i];
            delete pc[i];
        }
        return t;
    }

    Tensor<PlainText *> decrypt_tensor(const Tensor<CipherText *> &ct) {
        decrypted_ += ct.num_elements();
        if (threshold_ == 0) return cs_.decrypt_tensor(sk_, ct);
        Vector<Tensor<PartDecryptionResult *>> pdrs;
        for (const auto &sh : shares_) pdrs.push_back(cs_.part_decrypt_tensor(sh, ct));
        auto res = cs_.combine_part_decryption_results_tensor(ct, pdrs);
        for (auto &p : pdrs) {
            p.flatten();
            for (size_t i = 0; i < p.num_elements(); i++) delete p[i];
        }
        return res;
    }
    PlainText decrypt(const CipherText &ct) {
        Tensor<CipherText *> t(1, const_cast<CipherText *>(&ct));
        auto r = decrypt_tensor(t);
        PlainText out = *r[0];
        delete r[0];
        return out;
    }
    size_t decrypted_elements() const { return decrypted_; }

  private:
    CryptoSystem &cs_;
    SecretKey sk_;
    PublicKey pk_;
    size_t threshold_ = 0;
    Vector<SecretKeyShare> shares_;
    size_t decrypted_ = 0;
};

template <typename CryptoSystem, typename Client = LocalSMPCClient<CryptoSystem>>
class LocalCipherTextMultiplier {
  public:
    using CipherText = typename CryptoSystem::CipherText;
    using PlainText = typename CryptoSystem::PlainText;

    explicit LocalCipherTextMultiplier(Client &client) : client_m(client) {}

    CipherText multiply_ciphertexts(const CipherText &ct1, const CipherText &ct2) {
        Tensor<CipherText *> a(1, const_cast<CipherText *>(&ct1)), b(1, const_cast<CipherText *>(&ct2));
        auto r = handle_vector_ciphertext_mul(a, b);
        CipherText out = *r[0];
        delete r[0];
        return out;
    }

    // 0-D, 1-D (element-wise) and 2-D (matrix product) as in the reference (:40-112)
    Tensor<CipherText *> multiply_ciphertext_tensors(const Tensor<CipherText *> &ct1, const Tensor<CipherText *> &ct2) {
        if (ct1.is_zero_degree() && ct2.is_zero_degree())
            return Tensor<CipherText *>(new CipherText(multiply_ciphertexts(*ct1.get_value(), *ct2.get_value())));
        if (ct1.ndim() == 1) return handle_vector_ciphertext_mul(ct1, ct2);
        if (ct1.ndim() == 2) {
            // all n*m*p element products in one batch (:52-75), then one accumulation kernel
            // instead of the n*p*m serial nucomp loop (:85-101)
            const size_t n = ct1.shape()[0], m = ct1.shape()[1], p = ct2.shape()[1], nmp = n * m * p;
            if (ct2.shape()[0] != m) throw std::invalid_argument("Tensor shapes must be equal");
            Tensor<CipherText *> ct1_nmp(nmp, nullptr), ct2_nmp(nmp, nullptr);
            for (size_t i = 0; i < n; i++)
                for (size_t j = 0; j < m; j++)
                    for (size_t k = 0; k < p; k++) {
                        ct1_nmp[i * m * p + j * p + k] = ct1[i * m + j];
                        ct2_nmp[i * m * p + j * p + k] = ct2[j * p + k];
                    }
            auto res_nmp = handle_vector_ciphertext_mul(ct1_nmp, ct2_nmp);
            auto &cs = client_m.crypto_system();
            auto zero = cs.encrypt(client_m.network_public_key(), cs.make_plaintext(0));
            auto res = cs.accumulate_ciphertext_tensor(zero, res_nmp, n, m, p);
            for (size_t i = 0; i < nmp; i++) delete res_nmp[i];
            return res;
        }
        throw std::runtime_error("Not implemented");
    }

    // element-wise products of two 1-D ciphertext tensors (:115-160), same order of calls
    Tensor<CipherText *> handle_vector_ciphertext_mul(const Tensor<CipherText *> &ct1, const Tensor<CipherText *> &ct2) {
        const size_t n = ct1.shape()[0];
        if (ct2.num_elements() != n) throw std::invalid_argument("Tensor shapes must be equal");
        auto &cs = client_m.crypto_system();
        const auto &pk = client_m.network_public_key();
        auto triplets = client_m.get_beavers_triplets(n);
        Tensor<CipherText *> a_tensor(n, nullptr), b_tensor(n, nullptr), c_tensor(n, nullptr);
        for (size_t i = 0; i < n; i++) {
            a_tensor[i] = triplets.at(i, 0);
            b_tensor[i] = triplets.at(i, 1);
            c_tensor[i] = triplets.at(i, 2);
        }
        auto neg_a_tensor = cs.negate_ciphertext_tensor(pk, a_tensor);
        auto neg_b_tensor = cs.negate_ciphertext_tensor(pk, b_tensor);
        auto ct1_neg_a = cs.add_ciphertext_tensors(pk, ct1, neg_a_tensor);
        auto ct2_neg_b = cs.add_ciphertext_tensors(pk, ct2, neg_b_tensor);
        auto pt1 = client_m.decrypt_tensor(ct1_neg_a);
        auto pt2 = client_m.decrypt_tensor(ct2_neg_b);
        auto pt1_pt2 = cs.multiply_plaintext_tensors(pt1, pt2);
        auto enc_pt1_pt2 = cs.encrypt_tensor(pk, pt1_pt2);
        auto pt1_b = cs.scal_ciphertext_tensors(pk, pt1, b_tensor);
        auto pt2_a = cs.scal_ciphertext_tensors(pk, pt2, a_tensor);
        auto s1 = cs.add_ciphertext_tensors(pk, pt1_b, pt2_a);
        auto s2 = cs.add_ciphertext_tensors(pk, s1, c_tensor);
        auto ct = cs.add_ciphertext_tensors(pk, s2, enc_pt1_pt2);
        for (size_t i = 0; i < n; i++) {
            delete triplets.at(i, 0);
            delete triplets.at(i, 1);
            delete triplets.at(i, 2);
            delete neg_a_tensor[i];
            delete neg_b_tensor[i];
            delete ct1_neg_a[i];
            delete ct2_neg_b[i];
            delete pt1[i];
            delete pt2[i];
            delete pt1_pt2[i];
            delete enc_pt1_pt2[i];
            delete pt1_b[i];
            delete pt2_a[i];
            delete s1[i];          // the reference leaks these two intermediates (:151-153)
            delete s2[i];
        }
        return ct;
    }

  private:
    Client &client_m;
};

}  // namespace CoFHE
