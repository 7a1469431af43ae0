/* cofhe_hip.h -- C ABI of the MI355X evaluation engine for CoFHE's local ciphertext-tensor
 * path.  Plain pointers and sizes only; no C++, GMP or torch types cross this boundary.
 *
 * What each entry point replaces in the reference (paths under /root/reference):
 *   cofhe_hip_compose_records        the hot loop of CPUCryptoSystem::add_ciphertext_tensors,
 *                                    include/x86_64/cpu_cryptosystem_tensor_ops.inl:242-264
 *                                    (Cl_G.nucomp / Cl_Delta.nucomp per element)
 *   cofhe_hip_pow_records            the 1-D branch of scal_ciphertext_tensors,
 *                                    cpu_cryptosystem_tensor_ops.inl:316-338 (ClassGroup::nupow)
 *                                    and negate_ciphertext_tensor, :170-192
 *   cofhe_hip_scal_matmul_records    the 2-D branch, cpu_cryptosystem_tensor_ops.inl:342-461
 *                                    (qfi_nupow tables, include/x86_64/qfi.inl:1-135, fused
 *                                    with the accumulation loop :403-417)
 *   cofhe_hip_decrypt_records        decrypt_tensor's per-element work, cpu_cryptosystem_tensor_ops.inl:21-33
 *   cofhe_hip_part_decrypt_records,  part_decrypt_tensor / combine_part_decryption_results_tensor,
 *   cofhe_hip_combine_part_...       cpu_cryptosystem_tensor_ops.inl:35-73 (cpu_cryptosystem_distributed.inl:231-285)
 *   cofhe_hip_*_bytes                the same three operations on the reference's binary tensor
 *                                    format (serialize/deserialize_ciphertext_tensor,
 *                                    include/x86_64/cpu_cryptosystem.inl:320-508; plaintext
 *                                    tensors :229-318), i.e. what a Tensor<CipherText*> call
 *                                    site hands over after serialising
 *   cofhe_hip_bytes_to_records /     the (de)serialisers themselves, producing / consuming the
 *   cofhe_hip_records_to_bytes       device layout (cofhe_amd/csrc/layout.hpp)
 *
 * Records: one quadratic form = 168 little-endian u32 words (a[40] |b|[40] c[80] sign pad[7]);
 * one ciphertext = 2 records (c1, c2).  Exponents: EXP_WORDS u32 magnitude words + 1 sign
 * word each (cofhe_hip_exp_words()).
 *
 * Concurrency: a context may be shared by host threads: the entry points that use its grow-only workspace, its
 * cached tables or its status area (matrix product, encrypt, decrypt, part_decrypt, combine, accumulate, validate,
 * device_status, the *_bytes operations) take the context's lock for the duration of the call; their kernels run
 * in stream order, so callers that pass DIFFERENT streams for those operations must order them themselves (the
 * NULL stream, which HIPCryptoSystem uses, needs nothing).  compose / pow / the converters are stateless.
 *
 * All functions return 0 on success, a negative COFHE_HIP_E* code otherwise;
 * cofhe_hip_last_error() gives the message for the calling thread.  The library never falls
 * back to a CPU computation: without a usable GPU cofhe_hip_ctx_create fails.
 */
#ifndef COFHE_HIP_H
#define COFHE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COFHE_HIP_OK 0
#define COFHE_HIP_EINVAL (-1)   /* malformed buffer / value exceeds the limb capacity        */
#define COFHE_HIP_ESHAPE (-2)   /* "Tensor shapes must be equal" / "Vector sizes must be equal" */
#define COFHE_HIP_ENDIM (-3)    /* "Tensors must be 0D, 1D or 2D for now"                      */
#define COFHE_HIP_EHIP (-4)     /* HIP runtime error (message carries hipGetErrorString)       */
#define COFHE_HIP_ENOMEM (-5)

typedef struct cofhe_hip_ctx cofhe_hip_ctx;

const char *cofhe_hip_last_error(void);
int cofhe_hip_record_words(void);   /* 168 */
int cofhe_hip_exp_words(void);      /* magnitude words per exponent (sign word follows) */

/* Context bound to one GPU and one discriminant; absdelta = little-endian bytes of |Delta|. */
int cofhe_hip_ctx_create(int device, const uint8_t *absdelta_le, size_t len, cofhe_hip_ctx **out);
void cofhe_hip_ctx_destroy(cofhe_hip_ctx *ctx);

/* device memory (so a host language needs no HIP binding).  Freed blocks are kept by the context and handed out again
 * for the same size class (at most 12.5 % above the request): cofhe_hip_free does not synchronise the device as hipFree does -- the block is
 * reused only after everything that was submitted to the null stream or a blocking stream before the free has run.
 * Buffers last used on a NON-blocking stream (PyTorch's pool streams are) are freed with cofhe_hip_free_on_stream,
 * which orders the reuse after the work queued on that stream.  When the device runs out of memory (here, for the
 * context's workspace or its tables) the cache is released and the allocation retried.
 * cofhe_hip_trim(ctx, keep) sets the cache limit (default: an eighth of the device memory, at most 64 GiB) and
 * releases the cache if it holds more. */
int cofhe_hip_malloc(cofhe_hip_ctx *ctx, size_t bytes, void **dptr);
int cofhe_hip_free(cofhe_hip_ctx *ctx, void *dptr);
int cofhe_hip_free_on_stream(cofhe_hip_ctx *ctx, void *dptr, void *stream);
int cofhe_hip_trim(cofhe_hip_ctx *ctx, size_t keep_bytes);
/* Launcher decisions of the matrix product that a caller may pin (0 = automatic, the default):
 *   "wnaf_width"       2..8: window width of the exponent recoding (automatic: minimises table + chain work)
 *   "ladder_form"      shared-exponent ladders (decryption): 0 by their number, 1 a pair of wavefronts per ladder (wide
 *                      layout: one squares, one multiplies), 2 the 8-lane in-wave form (<= 8 ladders), 3 the throughput
 *                      kernel, 4 one wavefront per ladder (wide layout, left to right with a table)
 *   "matmul_tree"      -1: the launcher decides; 1: the matrix product as per-position product trees + a Horner chain;
 *                      0: lockstep chains (the form of rounds 1-3)
 *   "matmul_segments"  >= 1: pieces the inner dimension is cut into when the product has few outputs
 *   "profile_kernels"  != 0: cofhe_hip_scal_matmul_records brackets each of its kernels with HIP events on the launch
 *                      stream; cofhe_hip_profile_read(ctx, "k_tree_level" | "k_scal_matmul_wnaf" | "k_pow_table" | "k_wnaf_digits", ...)
 *                      waits for them and returns the summed duration and the launch count (clear != 0 drops all spans)
 * The results do not depend on them; tests pin them to drive every width through the parity checker. */
int cofhe_hip_ctx_set_option(cofhe_hip_ctx *ctx, const char *name, int64_t value);
int cofhe_hip_profile_read(cofhe_hip_ctx *ctx, const char *kernel, float *total_ms, uint32_t *launches, int clear);
int cofhe_hip_upload(cofhe_hip_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes, void *stream);
int cofhe_hip_download(cofhe_hip_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes, void *stream);
int cofhe_hip_stream_sync(cofhe_hip_ctx *ctx, void *stream);

/* Device status word: every data-dependent loop of the kernels has a trip-count cap; a cap that is hit (possible only
 * for records that are not reduced forms of the context's discriminant) sets a bit instead of hanging or passing
 * silently: 1 = remainder sequence, 2 = reduction, 4 = division (by zero / no end).  clear != 0 resets it.
 * Synchronises `stream`. */
int cofhe_hip_device_status(cofhe_hip_ctx *ctx, uint32_t *word, int clear, void *stream);
/* *all_valid = 1 when every one of the n form records is a reduced form of the context's discriminant
 * (a, c > 0, |b| <= a <= c, canonical sign, b^2 - 4ac = Delta).  cofhe_hip_unpack_tensor_device and the *_bytes
 * operations run this on everything they deserialise; records produced by the kernels themselves need no check.
 * Synchronises `stream`. */
int cofhe_hip_validate_records(cofhe_hip_ctx *ctx, const void *d_records, uint64_t n_records, int *all_valid, void *stream);

/* ---- kernels on device-resident records (stream: hipStream_t, NULL = default stream) ---- */
/* out[i] = a[i] o b[i] for n_records forms (a ciphertext tensor of E elements is 2E records) */
int cofhe_hip_compose_records(cofhe_hip_ctx *ctx, const void *d_a, const void *d_b, void *d_out,
                              uint64_t n_records, void *stream);
/* out[i] = a[i] o b[i], ONE composition per wavefront in the wavefront-wide layout (csrc/wide.hpp: two limbs per lane over the
 * 64 lanes, no workgroup protocol): the composition of the latency kernels -- the ladders of cofhe_hip_decrypt_records /
 * cofhe_hip_part_decrypt_records when there are few of them -- as an entry of its own, for parity tests and for timing one
 * composition by itself (reps > 1 repeats it inside the kernel).  Same result records as cofhe_hip_compose_records.
 * *fallbacks (optional): how many pairs the wide route declined and took through the 8-lane code (synchronises). */
int cofhe_hip_compose_wide_records(cofhe_hip_ctx *ctx, const void *d_a, const void *d_b, void *d_out, uint64_t n_records,
                                   uint32_t reps, uint32_t *fallbacks, void *stream);
/* out[i] = a[i] + b[i] for n_ct ciphertexts (2 records each; same result records as cofhe_hip_compose_records over
 * 2 n_ct records).  When every ciphertext of a shares one c1 and every ciphertext of b shares one c1 -- tensors that
 * encrypt_tensor made with its one r per tensor (cpu_cryptosystem_tensor_ops.inl:7-12), and sums of such tensors --
 * the composition c1 o c1' is computed once and copied: n_ct + 1 compositions instead of 2 n_ct.  Detected on the
 * device per call (one pass over the c1 records); tensors with differing c1 take the plain path.  d_out may be d_a. */
int cofhe_hip_add_ciphertext_records(cofhe_hip_ctx *ctx, const void *d_a, const void *d_b, void *d_out, uint64_t n_ct, void *stream);
/* out[2e+h] = base[2e+h] ^ exp[e] for E ciphertexts (h = 0,1) */
int cofhe_hip_pow_records(cofhe_hip_ctx *ctx, const void *d_base, const void *d_exp, void *d_out,
                          uint64_t n_ciphertexts, void *stream);
/* out[i,k] = zero o prod_j cts[i,j]^s[j,k];  cts n x m, s m x p (exponent records), zero 1 ct.
 * NOT purely stream-ordered: the call synchronises `stream` once or twice before it returns its last launches -- the
 * window width follows the longest exponent (a 4-byte read-back) and the product tree is sized by per-level totals (a
 * ~100-byte read-back); the launches that follow are asynchronous as everywhere else. */
int cofhe_hip_scal_matmul_records(cofhe_hip_ctx *ctx, const void *d_cts, const void *d_exp,
                                  const void *d_zero, void *d_out, uint32_t n, uint32_t m, uint32_t p,
                                  void *stream);
/* decryption: for each of n ciphertexts, m with c2 o (c1^sk)^-1 = f^m.  sk: one exponent record on
 * the device; f_record: HOST pointer to the 168-word record of f = (2^(2k), 2^(k+1), 1 - Delta_K)
 * (its table of f^(-2^j) is built on first use and cached in the context).  d_out receives
 * ceil(k/32) little-endian words of m followed by one status word (0 ok, 1 = not in <f>) per
 * ciphertext.  Reference: CL_HSM2k::decrypt via cpu_cryptosystem_tensor_ops.inl:21-33.
 * When all n >= 64 ciphertexts carry the same c1 (a tensor made by encrypt_tensor, or a sum of such tensors) c1^sk is
 * computed once and copied; tensors with differing c1 run one ladder per ciphertext.  Same for part_decrypt below. */
int cofhe_hip_decrypt_records(cofhe_hip_ctx *ctx, const void *d_cts, const void *d_sk, const uint32_t *f_record,
                              void *d_out, uint64_t n_ciphertexts, uint32_t kbits, void *stream);
/* out[i,k] = zero o prod_j x[i,j,k]: x is n x m x p ciphertexts (row-major), zero 1 ciphertext, out n x p.
 * The accumulation loop of the ciphertext x ciphertext matrix product,
 * include/smpc/ciphertext_multiplications.hpp:85-101. */
int cofhe_hip_accumulate_records(cofhe_hip_ctx *ctx, const void *d_x, const void *d_zero, void *d_out, uint32_t n,
                                 uint32_t m, uint32_t p, void *stream);
/* out[i] = base[i] ^ exp[i] on single forms (n_forms records, n_forms exponent records) */
int cofhe_hip_pow_form_records(cofhe_hip_ctx *ctx, const void *d_base, const void *d_exp, void *d_out,
                               uint64_t n_forms, void *stream);
/* out = base^e for a base that recurs (h of the cryptosystem, a public key): base_record and exp_record are HOST
 * pointers (168 / 32 words), d_out one form record on the device.  The context keeps the table base^(2^j) of the last
 * few bases (first use of a base: one chain of ~1000 squarings, ~0.5 s); afterwards the power is a product tree over
 * the ~bits/3 entries the signed binary digits of e select -- a few milliseconds.  Reference: h^r and pk^r of
 * encrypt_tensor, cpu_cryptosystem_tensor_ops.inl:7-12; h^sk of keygen, cpu_cryptosystem.inl:6-9. */
int cofhe_hip_pow_fixed_base_record(cofhe_hip_ctx *ctx, const uint32_t *base_record, const uint32_t *exp_record, void *d_out,
                                    void *stream);
/* n <= 4 such powers at once (base_records: n x 168 words, exp_records: n x 32 words, both HOST; d_out: n records):
 * one gather and one product tree for all of them, so h^r and pk^r of an encryption cost one tree's latency. */
int cofhe_hip_pow_fixed_base_records(cofhe_hip_ctx *ctx, uint32_t n, const uint32_t *base_records, const uint32_t *exp_records,
                                     void *d_out, void *stream);
/* encryption with given randomness: out[e] = (c1, pk^r o f^(m_e mod 2^k)).  d_plain: n exponent records
 * (plaintexts, sign honoured); d_c1_pkr: 2 form records on the device, c1 = h^r then pk^r (two powers the
 * caller takes once per tensor with cofhe_hip_pow_form_records); f_record as for decryption (the same cached
 * table of f^(-2^j) serves as the fixed-base table).  Reference: encrypt_tensor,
 * cpu_cryptosystem_tensor_ops.inl:1-19 (one r per tensor, element i = CipherText(hsm2k, m_i, c1, pkr)). */
int cofhe_hip_encrypt_records(cofhe_hip_ctx *ctx, const void *d_plain, const void *d_c1_pkr, const uint32_t *f_record,
                              void *d_out, uint64_t n_ciphertexts, uint32_t kbits, void *stream);
/* threshold decryption, party side: out[e] = c1[e] ^ share (one form record per ciphertext; d_share: one
 * exponent record on the device).  Reference: partDecrypt, cpu_cryptosystem_distributed.inl:259-269, looped
 * by part_decrypt_tensor, cpu_cryptosystem_tensor_ops.inl:35-48. */
int cofhe_hip_part_decrypt_records(cofhe_hip_ctx *ctx, const void *d_cts, const void *d_share, void *d_out,
                                   uint64_t n_ciphertexts, void *stream);
/* threshold decryption, combiner side: m with c2 o (prod_i parts[i][e]^lambda[i])^-1 = f^m.  d_parts:
 * n_parts x n_ciphertexts form records, party-major; lambda: HOST array of n_parts coefficients, each +1
 * or -1 (the reference's compute_lambda gives (1, -1, ..., -1)); n_parts <= 64.  Output as
 * cofhe_hip_decrypt_records.  Reference: finalDecrypt / compute_d, cpu_cryptosystem_distributed.inl:231-285,
 * looped by combine_part_decryption_results_tensor, cpu_cryptosystem_tensor_ops.inl:50-73. */
int cofhe_hip_combine_part_decryptions_records(cofhe_hip_ctx *ctx, const void *d_cts, const void *d_parts,
                                               uint32_t n_parts, const int32_t *lambda, const uint32_t *f_record,
                                               void *d_out, uint64_t n_ciphertexts, uint32_t kbits, void *stream);
/* the same compose launch repeated `iters` times between two HIP events on `stream`;
 * *ms_per_launch = elapsed / iters (used by bench.py for the roofline figure) */
int cofhe_hip_time_compose(cofhe_hip_ctx *ctx, const void *d_a, const void *d_b, void *d_out,
                           uint64_t n_records, int iters, void *stream, float *ms_per_launch);

/* How an entry point would carve the context's workspace (host only, no GPU, no context): the launchers take their
 * regions from the same plan functions, so a CPU test can check every offset and size for any operand count.
 *   op "pow_shared"       args: n_ladders, front_bytes        (k_pow_shared: part_decrypt / decrypt's c1^sk)
 *      "decrypt"          args: n_ct, shared_c1 (0 / 1)       (cofhe_hip_decrypt_records: front = one record per ciphertext)
 *      "part_decrypt"     args: n_ct, shared_c1               (cofhe_hip_part_decrypt_records: no front)
 *      "scal_matmul"      args: n, m, p, exp_bits, w, segs
 *      "accumulate_tree"  args: n, m, p
 *      "encrypt_chunk"    args: n_elements, kbits
 *      "fixed_base"       args: n_powers, max_entries
 * regions[i] = name, byte offset, byte count, in workspace order; *total_bytes = what ensure_workspace is asked for. */
typedef struct {
    char name[24];
    uint64_t offset, bytes;
} cofhe_hip_ws_region;
int cofhe_hip_workspace_plan(const char *op, const uint64_t *args, uint32_t n_args, cofhe_hip_ws_region *regions,
                             uint32_t cap, uint32_t *n_regions, uint64_t *total_bytes);

/* Stream timer: two HIP events on `stream` -- the stream the library's kernels are launched on (a torch.cuda.Event only
 * sees torch's current stream).  start records the first event; stop records the second, waits for it, returns the time
 * between the two in *ms and releases the timer.  bench.py times every secondary figure of its JSON line with this. */
int cofhe_hip_timer_start(cofhe_hip_ctx *ctx, void *stream, void **timer);
int cofhe_hip_timer_stop(cofhe_hip_ctx *ctx, void *timer, void *stream, float *ms);

/* ---- binary tensor format <-> records (host side, no GPU work) ---- */
/* ciphertext tensor bytes -> malloc'd record array (free with cofhe_hip_host_free) */
int cofhe_hip_bytes_to_records(const uint8_t *bytes, size_t len, uint32_t *ndim, uint32_t shape[8],
                               uint32_t **records, uint64_t *n_records);
int cofhe_hip_records_to_bytes(const uint32_t *records, uint64_t n_records, uint32_t ndim,
                               const uint32_t *shape, uint8_t **bytes, size_t *len);
/* partial-decryption tensors (one form per element; serialize_part_decryption_result_tensor,
 * cpu_cryptosystem.inl:510-559 / :561-635): the same layout with 3 integers per element */
int cofhe_hip_pdr_bytes_to_records(const uint8_t *bytes, size_t len, uint32_t *ndim, uint32_t shape[8],
                                   uint32_t **records, uint64_t *n_records);
int cofhe_hip_pdr_records_to_bytes(const uint32_t *records, uint64_t n_records, uint32_t ndim,
                                   const uint32_t *shape, uint8_t **bytes, size_t *len);
/* ---- the same formats produced / consumed on the GPU (cofhe_amd/csrc/wire.hip) ----
 * kind: 2 = ciphertext tensor (2 form records per element), 1 = partial-decryption tensor (1 form record),
 * 0 = plaintext tensor (1 exponent record).  d_bytes holds the serialised tensor in device memory (what
 * a caller uploads verbatim from the socket / file); records come out in the layout the kernels use.
 * Both calls synchronise `stream` (the header and the error / length words travel back to the host). */
int cofhe_hip_unpack_tensor_device(cofhe_hip_ctx *ctx, const void *d_bytes, size_t len, int kind, void *d_records,
                                   uint64_t capacity_records, uint32_t *ndim, uint32_t shape[8], uint64_t *n_records,
                                   void *stream);
int cofhe_hip_pack_tensor_device(cofhe_hip_ctx *ctx, const void *d_records, uint64_t n_records, int kind, uint32_t ndim,
                                 const uint32_t *shape, void *d_bytes, size_t capacity, size_t *len, void *stream);
/* upper bound of the serialised size, for sizing d_bytes */
size_t cofhe_hip_packed_size_bound(uint64_t n_records, int kind, uint32_t ndim);
/* plaintext tensor bytes -> exponent records */
int cofhe_hip_bytes_to_exponents(const uint8_t *bytes, size_t len, uint32_t *ndim, uint32_t shape[8],
                                 uint32_t **exps, uint64_t *n_exps);
void cofhe_hip_host_free(void *p);

/* ---- whole operations on host buffers in the reference's binary formats ---- */
int cofhe_hip_add_ciphertext_tensors_bytes(cofhe_hip_ctx *ctx, const uint8_t *t1, size_t l1,
                                           const uint8_t *t2, size_t l2, uint8_t **out, size_t *outlen);
/* s: plaintext tensor; 1-D x 1-D -> element-wise, 2-D x 2-D -> matmul (zero: 1-element tensor) */
int cofhe_hip_scal_ciphertext_tensors_bytes(cofhe_hip_ctx *ctx, const uint8_t *s, size_t ls,
                                            const uint8_t *cts, size_t lc, const uint8_t *zero, size_t lz,
                                            uint8_t **out, size_t *outlen);

/* ---- more than one GPU: one process (and one context) per GPU ------------------------------------------------------
 * The path shards by rows with no exchange between chained operations (row i of the result of
 * add_ciphertext_tensors / scal_ciphertext_tensors needs only row i of the ciphertext operand: reference loops
 * cpu_cryptosystem_tensor_ops.inl:242-264, :396-417; the plaintext matrix and Enc(0) are replicated).  The one
 * collective reassembles a row-sharded result: RCCL all-gather of the fixed-size records on the device buffers.
 * The reference has no counterpart (one tensor per compute node); these are additions for the host application.
 *
 * cofhe_hip_shard_rows: rank's contiguous row block [row0, row0 + n_local), remainder rows to the low ranks.
 * cofhe_hip_comm_unique_id: rank 0 draws the id and hands it to the other ranks over the application's own channel.
 * cofhe_hip_comm_create: collective over all ranks (ncclCommInitRank on the context's device).
 * cofhe_hip_all_gather_rows: d_local = this rank's rows (n_local * row_bytes bytes), d_out = all n_rows rows on every
 *   rank; row_bytes = columns * records per element * 672.  Runs on `stream` (the stream of the compute that produced
 *   d_local: the collective is ordered after it with no host synchronisation).  Equal blocks: one ncclAllGather; ragged
 *   blocks: one ncclBroadcast per non-empty block inside a group.
 * cofhe_hip_gather_plan: that decision as data (host only, no GPU, no RCCL): byte offset and byte count of every rank's
 *   block in the assembled tensor, *uniform = 1 for the ncclAllGather route.  offsets / counts: `world` entries each.
 * cofhe_hip_comm_info: world and rank the communicator was made with, and the rank count RCCL reports (ncclCommCount).
 * cofhe_hip_comm_set_option: "force_grouped_broadcast" != 0 takes the ragged route for equal blocks too (testing the
 *   grouped-broadcast branch where the row count happens to divide). */
typedef struct cofhe_hip_comm cofhe_hip_comm;
#define COFHE_HIP_COMM_ID_BYTES 128
void cofhe_hip_shard_rows(uint64_t n_rows, uint32_t world, uint32_t rank, uint64_t *row0, uint64_t *n_local);
int cofhe_hip_comm_unique_id(uint8_t id[COFHE_HIP_COMM_ID_BYTES]);
int cofhe_hip_comm_create(cofhe_hip_ctx *ctx, const uint8_t id[COFHE_HIP_COMM_ID_BYTES], uint32_t world, uint32_t rank,
                          cofhe_hip_comm **out);
void cofhe_hip_comm_destroy(cofhe_hip_comm *comm);
int cofhe_hip_all_gather_rows(cofhe_hip_ctx *ctx, cofhe_hip_comm *comm, const void *d_local, uint64_t n_rows, uint64_t row_bytes,
                              void *d_out, void *stream);
int cofhe_hip_gather_plan(uint64_t n_rows, uint64_t row_bytes, uint32_t world, uint64_t *offsets, uint64_t *counts, int *uniform);
int cofhe_hip_comm_info(cofhe_hip_comm *comm, uint32_t *world, uint32_t *rank, uint32_t *rccl_nranks);
int cofhe_hip_comm_set_option(cofhe_hip_comm *comm, const char *name, int64_t value);

#ifdef __cplusplus
}
#endif
#endif
