/*
 * glb.h — C ABI of libglb_hip.so, the MI355X (gfx950) implementation of genlm-backend's
 * autobatched next_token_logprobs hot path.
 *
 * Every entry point is `extern "C"`, takes plain device/host pointers and sizes, returns an
 * int status (GLB_OK == 0) and never throws.  On failure a thread-local message is available
 * through glb_last_error().  The library never allocates or frees caller memory: all buffers,
 * including scratch, are caller-owned (torch-allocated in the Python host) and must stay valid
 * until the passed HIP stream has reached the end of the call.  Kernels are enqueued on the
 * caller's stream; no entry point synchronises the device unless its comment says so.
 *
 * Each entry point cites the reference (genlm/genlm-backend) code it replaces; paths are
 * relative to the reference checkout.
 *
 * Arithmetic contract ("GLB math"): a logits row is a sequence of 4096-element chunks, each
 * with its own binary scale; the terms of a correctly-rounded-FMA polynomial exp are added in float32 in a fixed
 * order inside each of a chunk's 256 (lane, class) groups of 16 elements and as integers (floor(P * 2^36)) above
 * that, so logZ / lse / sampled token are bit-identical for any launch geometry, GPU count or particle split,
 * and are restated bit-for-bit by oracle/glb_oracle.c.
 */
#ifndef GLB_H
#define GLB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLB_ABI_VERSION 9

/* status codes */
enum {
  GLB_OK = 0,
  GLB_EINVAL = 1,      /* bad argument (null pointer, zero size, bad enum, misaligned ld) */
  GLB_EUNSUPPORTED = 2,/* valid request the build cannot serve (e.g. vocab too large for scratch) */
  GLB_EHIP = 3,        /* HIP runtime error (message carries hipGetErrorString) */
  GLB_ENOSPC = 4       /* caller-provided workspace too small */
};

/* element types of a logits matrix */
enum { GLB_F32 = 0, GLB_BF16 = 1, GLB_F16 = 2 };

/* mask kinds (README.md:57-70 builds two shared {0,-inf} masks; real grammars give per-row masks) */
enum {
  GLB_MASK_NONE = 0, /* no mask: logZ == 0 up to rounding, sample from the full distribution      */
  GLB_MASK_BITS = 1, /* table [n_masks, mask_ld] of uint32 words, bit j%32 of word j/32 == 1 ⇔
                        token j allowed (log-mask 0), 0 ⇔ forbidden (log-mask -inf).  The one-launch step
                        reads these rows as they are (every lane the words that hold its elements' bits:
                        what a grammar that changes every particle's mask every step hands over); the
                        two-launch forms bring them into GLB_MASK_PREPARED's layout in the workspace first */
  GLB_MASK_F32 = 2,  /* table [n_masks, mask_ld] of float additive log-masks (any value, -inf ok) */
  GLB_MASK_PREPARED = 3 /* bit masks already brought into the kernels' layout by glb_mask_prepare (same
                        meaning as GLB_MASK_BITS; saves one small launch per call for masks that are
                        built once, like the two README masks)                                     */
};

/* glb_step_args.flags.
 * GLB_STEP_HW_EXP: the second arithmetic contract of the step - a term is the hardware's 2^y,
 * t = v_exp_f32(fma(x, log2 e, -(N_c + 1))), instead of the polynomial.  Same chunks, same
 * (lane, class) float32 sums in load order, same integer sums, records and draws, so results are still deterministic and
 * independent of launch geometry and particle sharding ON gfx950; v_exp_f32 is within one ulp of 2^y (the polynomial:
 * 2.7e-6) but not correctly rounded, so oracle/glb_oracle.c restates this contract with exp2f and the comparison is by
 * tolerance: logZ / lse within 1e-4 of the reference (observed 1e-6), parity-mode tokens identical on every golden,
 * Philox tokens equal to the oracle's except where a draw lands within 2^-20 of a boundary of the inverse CDF
 * (tests/test_step_gpu.py counts them).  What it buys: the exponential is 9 of the 13 vector instructions per element of
 * the polynomial contract and 16-bit rows are bound by instruction issue (512 x 128256 bf16 31.5 ->
 * 27 us); cache.py:96 keeps the logits' dtype, so 16-bit rows are every Llama-class checkpoint's case.  float32 rows are
 * bound by memory and gain less (1024 x 50257: 38.0 -> 36.6 us): the Python host sets the flag for 16-bit rows by default
 * ("auto") and for float32 rows only when asked ("hw"), so that float32 results stay the oracle's bit for bit. */
enum { GLB_STEP_HW_EXP = 1 };

/* RNG modes of the categorical draw */
enum {
  GLB_RNG_NONE = 0,   /* no draw (out_token untouched)                                            */
  GLB_RNG_PHILOX = 1, /* in-kernel Philox4x32-10, two 64-bit uniforms per particle: exact integer inverse
                         CDF over the row's 4096-token chunks, then down the chunk's summation tree: lane,
                         class, element                                              */
  GLB_RNG_NOISE = 2   /* parity mode: caller supplies Exp(1) noise E[n_particles, noise_ld] drawn
                         the way torch.multinomial draws it on CPU; token = first argmax p_j/E_j
                         (README.md:87, base.py:137-141)                                          */
};

const char *glb_version(void);
int glb_abi_version(void);
/* copies the calling thread's last error message (NUL-terminated, truncated to n) */
int glb_last_error(char *buf, size_t n);
/* number of visible HIP devices, or -1 when the runtime cannot initialise (no GPU) */
int glb_device_count(void);

/*
 * Fused particle step.  Replaces, per particle,
 *     logps  = log_softmax(logits_row)                      cache.py:96
 *     masked = logps + mask                                 README.md:84
 *     logZ   = masked.logsumexp(-1)                         README.md:85
 *     token  = multinomial((masked - logZ).exp(), 1)        README.md:87   (base.py:136-141 with
 *                                                           logit_scale = 1/temperature)
 * in one pass over the logits row.  Particle i reads row row_of[i] (dedup fan-out of
 * hf.py:214-220,285-288) and mask row mask_id[i].  When the mask is a function of the context - as in
 * the README, where it depends on len(context) only - pass the ids per logits ROW (row_mask_id) instead:
 * a row shared by several particles is then reduced once and only the draw is done per particle (any number of
 * particles per row: the reduction leaves what every draw needs, whoever makes it).
 */
typedef struct glb_step_args {
  uint32_t struct_size; /* sizeof(glb_step_args) — ABI guard */
  /* logits of the unique contexts */
  const void *logits; /* [n_rows, ld] device */
  int32_t dtype;      /* GLB_F32 / GLB_BF16 / GLB_F16 */
  int64_t n_rows;
  int64_t vocab; /* V */
  int64_t ld;    /* row pitch in elements, >= vocab */
  float logit_scale; /* x' = fl(x * logit_scale) when != 1 (1/temperature of base.py:136) */
  /* particles */
  int64_t n_particles;
  const int32_t *row_of; /* [n_particles] device, nullable ⇒ identity (needs n_particles == n_rows) */
  /* mask */
  int32_t mask_kind;
  const void *mask;  /* [n_masks, mask_ld] device (uint32 words or float) */
  int64_t mask_ld;   /* pitch in words (BITS, >= ceil(V/32)) or floats (F32, >= V) */
  int64_t n_masks;
  const int32_t *mask_id; /* [n_particles] device, nullable ⇒ identity (needs n_masks == n_particles)
                             or, when n_masks == 1, everybody uses mask 0 */
  const int32_t *row_mask_id; /* [n_rows] device, nullable: mask ids per logits row instead of per particle
                             (exclusive with mask_id); null ids ⇒ identity over rows / mask 0 as above */
  /* rng */
  int32_t rng_mode;
  const float *noise; /* GLB_RNG_NOISE: [n_particles, noise_ld] device */
  int64_t noise_ld;   /* >= vocab, or 0: ONE row shared by every particle (base.py:148-179 seeds every sequence of a
                         batch_sample call alike) */
  uint64_t seed;          /* GLB_RNG_PHILOX key */
  uint64_t offset;        /* GLB_RNG_PHILOX counter word (e.g. SIS step number) */
  int64_t particle_base;  /* global index of particle 0 of this call (multi-GPU shards) */
  /* outputs, each nullable */
  float *out_logZ;   /* [n_particles] logsumexp(log_softmax(x)+mask) */
  float *out_lse;    /* [n_particles] logsumexp(x) of the particle's row */
  int32_t *out_token;/* [n_particles] sampled id, -1 when every token is masked out (-2, with NaN logZ / lse: a
                        finishing wave gave up waiting for its row's records - a failed launch, never a result) */
  float *out_margin; /* [n_particles] GLB_RNG_NOISE only: (winner - runner-up) / winner of the race e_j / E_j, i.e. how
                        far the draw is from a tie that float rounding could flip (1 if there is no runner-up) */
  int32_t flags;     /* 0 or GLB_STEP_HW_EXP; any other bit: GLB_EINVAL */
  /* device scratch of >= glb_step_workspace_bytes(...) bytes, 32-byte aligned: the per-chunk records (64 bytes per
     row and 4096-token chunk) the reducing waves hand to the per-particle waves (and, for GLB_MASK_BITS, the prepared
     masks).  See glb_workspace_init. */
  void *workspace;
  size_t workspace_bytes;
} glb_step_args;

size_t glb_step_workspace_bytes(int64_t n_particles, int64_t n_rows, int64_t vocab, int64_t n_masks);
int glb_logprob_mask_sample(const glb_step_args *args, void *hip_stream);
/*
 * The same call with two HIP events (hipEvent_t, created by the caller with timing enabled; either may be null) that
 * the step's first launch carries as its start stamp and its last launch as its stop stamp (hipExtLaunchKernel):
 * hipEventElapsedTime(start, stop) is then the duration of the step's launch(es) on the device - what a profiler
 * reports for them - without the marker packets of hipEventRecord around the call.  For measurement (bench.py).
 */
int glb_logprob_mask_sample_timed(const glb_step_args *args, void *hip_stream, void *start_event, void *stop_event);

/*
 * A step workspace that was zeroed and registered with glb_workspace_init is served in ONE launch: the waves that
 * reduce the rows tag every record with the call's epoch (counted per workspace by the library) and the waves that
 * finish the particles, dealt at the end of the same grid, sweep the records of their row until all tags are fresh.
 * Call it once after allocating the buffer (enqueues one memset on the stream), and glb_workspace_release before
 * freeing it or handing the memory to anything else.  Workspaces nobody initialised, calls made while the stream is
 * being captured into a graph, and GLB_RNG_NOISE calls take two launches - same results, bit for bit.  A workspace
 * serves one call at a time (as before: it is scratch).
 */
int glb_workspace_init(void *workspace, size_t workspace_bytes, void *hip_stream);
int glb_workspace_release(void *workspace);
/*
 * The one-launch forms (fused step, log-softmax rows) have waves that wait, inside the launch, for records other waves
 * of the same launch write.  HIP promises no dispatch order, so every such wait is bounded by a watchdog (2 s by
 * default; glb_set_spin_limit, 0 = default): a wave that gives up writes token -2 and NaN logZ / lse (NaN rows for
 * glb_log_softmax_rows) and adds 1 to the error word in the last 64 bytes of the registered workspace - which is why a
 * registered workspace serves the calls of ONE stream at a time, and why its size should come from the *_workspace_bytes
 * functions (they leave that room).  The calls themselves return GLB_OK: they do not synchronise.
 *   glb_workspace_check       synchronises `hip_stream`, returns GLB_EHIP (and clears the word) if any wave gave up since
 *                             the last check - the results of those calls must not be used -, GLB_OK otherwise
 *   glb_workspace_error_word  the word's device address, for hosts that already copy results back and want it to ride
 *                             along (null for a workspace that is not registered); nonzero = failed, as above
 * The reference's counterpart of "nobody is left with a silent wrong answer" is vllm.py:396-400.
 */
int glb_workspace_check(void *workspace, void *hip_stream);
const uint32_t *glb_workspace_error_word(void *workspace);
#define GLB_SPIN_NONE UINT64_MAX /* give up at the first poll that finds a record missing: forces the failure path (tests) */
int glb_set_spin_limit(uint64_t microseconds);

/*
 * Bring GLB_MASK_BITS rows into the layout the kernels read ([mask][chunk][vector][component] 64-bit lane
 * words) once, for masks that do not change between steps (the two README masks, README.md:57-70).  `dtype` is the element type of the
 * logits the masks will be used with (the lane layout differs between 4- and 2-byte elements).  Pass the
 * result as `mask` with mask_kind = GLB_MASK_PREPARED and the same n_masks / vocab.
 */
size_t glb_mask_prepared_bytes(int64_t n_masks, int64_t vocab);
int glb_mask_prepare(const uint32_t *mask_bits, int64_t n_masks, int64_t vocab, int64_t mask_ld, int32_t dtype,
                     void *out_prepared, size_t out_bytes, void *hip_stream);
/*
 * Masks that change a few at a time (a grammar moves one particle's mask per token, README.md:57-70 generalised): re-prepare
 * ONLY the masks rows[0 .. n_rows) (device int32; entries outside [0, n_masks) are skipped) of a buffer glb_mask_prepare
 * filled before - the other masks' lane words stay as they are, so a steady-state call pays for what changed, not for
 * n_masks x ceil(V / 8) bytes in and out.
 */
int glb_mask_prepare_rows(const uint32_t *mask_bits, int64_t n_masks, int64_t vocab, int64_t mask_ld, int32_t dtype,
                          const int32_t *rows, int64_t n_rows, void *prepared, size_t prepared_bytes, void *hip_stream);

/*
 * Materialise log-probabilities: out[r, j] = x[r, j] - logsumexp(x[r, :]).  Replaces the
 * per-position torch.log_softmax of TokenTrie.extend_cache (cache.py:93-98) and
 * next_token_logprobs_uncached (hf.py:422).  out_lse is optional.  workspace: device scratch of at least
 * glb_log_softmax_workspace_bytes(n_rows, vocab) bytes, 32-byte aligned.  On a workspace glb_workspace_init has seen
 * (and outside stream capture, rows of at most 262144 elements) the call is one launch that reads every logit once
 * and keeps it on the chip until its log-probability is written; otherwise three launches.  Same bits either way.
 * out_dtype: GLB_F32, or the logits' own dtype - the reference returns log-probabilities in the model's dtype
 * (cache.py:96 keeps it): the float32 result rounded to nearest even, out_ld in elements of that type; for 16-bit
 * models that is a third fewer bytes per call (512 x 128256 bf16: 263 MB instead of 394 MB).
 */
size_t glb_log_softmax_workspace_bytes(int64_t n_rows, int64_t vocab);
int glb_log_softmax_rows(const void *logits, int32_t dtype, int64_t n_rows, int64_t vocab,
                         int64_t ld, float logit_scale, void *out_logprobs, int32_t out_dtype, int64_t out_ld,
                         float *out_lse, void *workspace, size_t workspace_bytes, void *hip_stream);

/*
 * Convert {0,-inf}-style float log-masks (README.md:60-66, `.log()` of a 0/1 tensor) to the
 * packed GLB_MASK_BITS form: bit = (mask[j] == 0).  Values other than 0 / -inf set *out_nonbinary
 * (device int32, optional) to 1 so the caller can fall back to GLB_MASK_F32.
 */
int glb_mask_f32_to_bits(const float *mask, int64_t n_masks, int64_t vocab, int64_t mask_ld,
                         uint32_t *out_bits, int64_t bits_ld, int32_t *out_nonbinary,
                         void *hip_stream);

/*
 * Ragged contexts are passed as (tokens, starts, lengths): context i is
 * tokens[starts[i] .. starts[i] + lengths[i]), int32 token ids, int64 starts, int32 lengths, all on
 * the device.  This covers both a CSR buffer and the rows of a padded [n, cap] particle matrix.
 *
 * Context grouping: exact dedup of ragged token contexts, first-appearance order.  Replaces the
 * dict keyed by tuple(prompt) of batch_evaluate_queries (hf.py:214-220).
 *   out_group_of [n]   group id of every context (ids numbered by first appearance)
 *   out_rep      [n]   out_rep[g] = smallest context index in group g (only [0, n_groups) valid)
 *   out_n_groups [1]   device int32
 * workspace: device scratch of at least glb_group_contexts_workspace(n) bytes.
 * ctx_hashes (nullable, [n] device): the contexts' hashes as glb_hash_contexts makes them and glb_particles_advance
 * keeps them up to date - a context's hash extends token by token, so a loop that only appends never reads its
 * contexts again to group them (the grouping itself is by exact comparison: hashes only place the slots).
 */
size_t glb_group_contexts_workspace(int64_t n);
int glb_group_contexts(const int32_t *tokens, const int64_t *starts, const int32_t *lengths,
                       int64_t n, const uint64_t *ctx_hashes, int32_t *out_group_of, int32_t *out_rep,
                       int32_t *out_n_groups, void *workspace, size_t workspace_bytes, void *hip_stream);
int glb_hash_contexts(const int32_t *tokens, const int64_t *starts, const int32_t *lengths, int64_t n,
                      uint64_t *out_hashes, void *hip_stream);

/*
 * Cached-prefix match ("trie grouping"): for every context pick the longest cached prefix that
 * is a proper prefix of it — the deepest KV-bearing trie node walk_cache returns (hf.py:314-344).
 *   cached prefixes in the same (tokens, starts, lengths) form
 *   out_prefix [n] int32 index of the chosen cached prefix or -1; out_base [n] int32 its length
 *   (0 when none).  Ties on length resolve to the lowest prefix index.
 */
int glb_match_prefixes(const int32_t *tokens, const int64_t *starts, const int32_t *lengths,
                       int64_t n, const int32_t *prefix_tokens, const int64_t *prefix_starts,
                       const int32_t *prefix_lengths, int64_t n_prefixes, int32_t *out_prefix,
                       int32_t *out_base, void *hip_stream);

/*
 * Ragged -> padded gather for the batched prefill.  Replaces Query.prompt_padded /
 * attention_mask / position_ids and the three torch.tensor(...) builds of hf.py:55-70,232-246.
 * For each selected context s = sel[u] (u < n_sel; sel nullable ⇒ identity) with cached-prefix
 * length base[s] (nullable ⇒ 0):
 *   input_ids[u, t]      = tokens[starts[s] + base + t]         t <  len          else pad_id
 *   position_ids[u, t]   = base + t                             t <  len          else 0
 *   attention_mask[u, p] = 1 for p < base; 0 for base <= p < p_max;
 *                          1 for p_max <= p < p_max + len; 0 afterwards           (hf.py:58-64)
 *   last_index[u]        = len - 1        (row of the next-token logits, optional output)
 * with len = lengths[s] - base.  Outputs are int64 [n_sel, l_max] /
 * [n_sel, p_max + l_max], matching what transformers expects.
 */
int glb_gather_padded(const int32_t *tokens, const int64_t *starts, const int32_t *lengths,
                      const int32_t *sel, int64_t n_sel, const int32_t *base, int64_t pad_id, int64_t p_max,
                      int64_t l_max, int64_t *out_input_ids, int64_t *out_attention_mask,
                      int64_t *out_position_ids, int32_t *out_last_index, void *hip_stream);

/*
 * Batched prefix-KV assembly.  Replaces Query.past_padded + the per-layer torch.cat of
 * hf.py:33-53,247-271: out[u, h, p, :] = slab_k[h, p, :] for p < plen_k, zero up to p_max,
 * where k = prefix_of[u] (k < 0 ⇒ all zeros).  `slabs` is a device array of n_prefixes device
 * pointers to [heads, plen_k, head_dim] tensors of elem_bytes-wide elements (contiguous).
 */
int glb_gather_kv_padded(const void *const *slabs, const int32_t *slab_len, int64_t n_prefixes,
                         const int32_t *prefix_of, int64_t n_rows, int64_t heads,
                         int64_t head_dim, int64_t p_max, int32_t elem_bytes, void *out,
                         void *hip_stream);

/*
 * Particle bookkeeping of the SIS loop (README.md:82-91): for every particle that is active,
 *   log_weight += logZ; if token == eos_id (or token < 0) active = 0
 *   else tokens[i, length[i]++] = token; and a context that reaches max_len is deactivated.
 * `contexts` is the padded [n, ctx_ld] int32 matrix the ragged views are cut from.  ctx_hashes (nullable, [n]): the
 * contexts' hashes (glb_hash_contexts), extended by the appended token.
 */
int glb_particles_advance(int32_t *contexts, int64_t ctx_ld, int32_t *lengths, int32_t *active,
                          float *log_weights, const float *logZ, const int32_t *token, int64_t n,
                          int32_t eos_id, int32_t max_len, uint64_t *ctx_hashes, void *hip_stream);

/*
 * Weight normalisation over the full particle population (README.md:108-110) on the all-gathered
 * log-weight vector: out_probs = exp(lw - logsumexp(lw)); out_stats = {logsumexp(lw), ESS}.
 * Deterministic (fixed-point sums), so every rank of a sharded run computes identical values.
 */
int glb_normalize_weights(const float *log_weights, int64_t n, float *out_probs, float *out_stats,
                          void *hip_stream);

/*
 * Device-resident particle state (SURVEY.md §8 f1 / e): per-particle KV in preallocated slabs
 * [n_rows, heads, cap, head_dim] (one per layer and K/V) instead of the reference's per-query tuples that are
 * zero-padded and concatenated every batch (hf.py:33-53,247-271) or per-token trie slices (cache.py:103-191).
 *
 * glb_kv_append:      slab[row_of[i], h, pos[i], :] = new_rows[i, h, :]   - the new token's K or V of every forward
 *                     row (row_of nullable: row i; with rows SHARED by particles of equal contexts it is the block
 *                     table's row of forward row i); new_rows is addressed by element strides (usually a transposed view).
 * glb_kv_gather_rows: dst[t][i, h, p, :] = src[t][src_row_of[i], h, p, :] for p < len_of[i], all n_tensors
 *                     (layer, K|V) slabs in one launch through device arrays of device pointers.  Fans the KV of
 *                     the distinct prompts out to the particles and gathers ancestors' KV after a resampling
 *                     step; src_row_of[i] < 0 leaves row i untouched.  src and dst must not alias.
 * glb_gather_rows_i32: dst[i, :width] = src[row_of[i], :width]   - the particles' token matrices.
 */
int glb_kv_append(void *slab, const void *new_rows, const int32_t *pos, const int32_t *row_of, int64_t n_rows,
                  int64_t heads, int64_t cap, int64_t head_dim, int64_t new_stride_row, int64_t new_stride_head,
                  int32_t elem_bytes, void *hip_stream);
int glb_kv_gather_rows(const void *const *src, void *const *dst, int64_t n_tensors, int64_t n_rows, int64_t heads,
                       int64_t head_dim, int64_t src_cap, int64_t dst_cap, const int32_t *src_row_of,
                       const int32_t *len_of, int32_t elem_bytes, void *hip_stream);
int glb_gather_rows_i32(const int32_t *src, int64_t src_ld, const int32_t *row_of, int64_t n, int64_t width,
                        int32_t *dst, int64_t dst_ld, void *hip_stream);

/*
 * KV rows shared between contexts, decided on the device.  The reference keeps per-token KV on trie
 * nodes (cache.py:103-191, mlx.py:177-318); here the prefixes sit in slab rows and a block table says which.
 *
 * glb_match_rows: for every dedup group g < *n_groups (its context = rep[g]), the table row that holds exactly the
 *   context, else the row that holds its first L - 1 tokens, else -1 (the smallest row index wins among equals).  The
 *   table is (row_tok [n_rows, cap] int32, row_len [n_rows] - 0: the row holds nothing -, row_hash [n_rows]: the hash of
 *   glb_hash_contexts); hashes only select candidates, every candidate's tokens are compared.  out_hash [n] (nullable): the
 *   groups' own hashes, for the table update of glb_kv_plan.  Contexts longer than cap match nothing.
 *
 * glb_kv_plan: the block table of one step, one small launch.  In: the dedup of the step's contexts (group_of [n], rep,
 *   n_groups: glb_group_contexts' outputs), the contexts' lengths, and where every group's prefix sits now - old_row[g]
 *   (old_row_by_context == 0: glb_match_rows' output) or old_row[rep[g]] (old_row_by_context != 0: the previous step's
 *   out_row_of_context: a population that only appends).  Decides: the first group (by id) of every live row KEEPS it and
 *   appends in place; the other groups that grew out of that row get a COPY of the prefix in a free row (copy-on-append),
 *   groups without a row whose context fits one get a free row to be filled from an ENCODING, in that order, for as long as
 *   free rows last - handed out by ascending row index, or longest unused first when row_stamps [n_rows] (int64, the call
 *   that used the row last; updated to call_no for every row in use) is given.  Out (int32, device):
 *     out_group_row [n]       the group's row from now on (-1: none)
 *     out_logits_row [n]      the forward row of the group: groups with a prefix in a row first, the ones to encode behind
 *     out_rows_a / out_ctx_a / out_pos_a [n]   per forward row k < head[1]: slab row, context index, position of the token fed
 *     out_ctx_b / out_rows_b [n]               per encoded row k < head[2]: context index, the row that keeps its KV (-1: none)
 *     out_copy_src / out_copy_len [n_rows]     row r takes out_copy_len[r] positions of row out_copy_src[r] (-1: nothing)
 *     out_ctx_of_row / out_pos_of_row [n_rows] for a forward that runs on ALL rows where they lie: the context that feeds
 *                                              row r and the position its token goes to (-1: the row is not in this
 *                                              forward; -2: it is being filled from an encoding)
 *     out_row_of_context [n]  (nullable) out_group_row[group_of[i]]: next step's old_row
 *     out_head [8]            n_groups, rows with a prefix, rows to encode, copies, encoded rows nobody keeps, the longest
 *                             context to encode, free rows before the call, 0 - all the host has to read
 *   With row_tok != null the table rows of every group that holds a row now are rewritten (tokens zero-padded to cap,
 *   length, group_hash[g]).  workspace: glb_kv_plan_workspace(n, n_rows) bytes, 4-byte aligned.
  *   A group whose context has outgrown a row (length > cap) holds no row from this step on, whatever old_row says: it is
 *   encoded from its tokens and not kept, and the row it sat in is free again.  A position outside [0, cap) handed to
 *   glb_slab_attention appends nothing and gives NaN outputs for that row.
 */
int glb_match_rows(const int32_t *tokens, const int64_t *starts, const int32_t *lengths, const int32_t *rep,
                   const int32_t *n_groups, int64_t n, const int32_t *row_tok, const int32_t *row_len,
                   const uint64_t *row_hash, int64_t n_rows, int64_t cap, int32_t *out_old_row, uint64_t *out_hash,
                   void *hip_stream);
typedef struct glb_kv_plan_args {
  uint32_t struct_size; /* sizeof(glb_kv_plan_args) - ABI guard */
  int64_t n, n_rows, cap;
  const int32_t *group_of, *rep, *n_groups;
  const int32_t *old_row;
  int32_t old_row_by_context;
  const int32_t *lengths;
  int64_t *row_stamps; /* nullable */
  int64_t call_no;
  /* the table of what the rows hold (all nullable together) */
  int32_t *row_tok, *row_len;
  uint64_t *row_hash;
  const uint64_t *group_hash;
  const int32_t *tokens;
  const int64_t *starts;
  /* outputs */
  int32_t *out_group_row, *out_logits_row, *out_rows_a, *out_ctx_a, *out_pos_a, *out_ctx_b, *out_rows_b, *out_copy_src,
      *out_copy_len, *out_ctx_of_row, *out_pos_of_row, *out_row_of_context, *out_head;
  void *workspace;
  size_t workspace_bytes;
} glb_kv_plan_args;
size_t glb_kv_plan_workspace(int64_t n, int64_t n_rows);
int glb_kv_plan(const glb_kv_plan_args *args, void *hip_stream);

/*
 * Attention of a one-token forward over KV slab rows where they lie (the read side of the device-resident KV, SURVEY.md
 * §8 f1; the reference hands zero-padded per-query KV to the model's own attention, hf.py:247-281): for every row r and
 * query head h, softmax(q . K[r, h / G, 0 .. pos[r]] * scale) . V[r, h / G, 0 .. pos[r]], where position pos[r] is the token
 * of this forward - its K / V (k_new, v_new: [n_rows, kv_heads, head_dim], each by its own element strides: the projections' outputs)
 * are used from where they are AND written to slab position pos[r] (glb_kv_append, fused).  q by element strides
 * (row, head), unit inner stride; slabs [n_rows, kv_heads, cap, head_dim] contiguous; out [n_rows, heads, head_dim]
 * contiguous; all of one dtype, float32 accumulation; head_dim 16 / 32 / 64 / 128 (GLB_EUNSUPPORTED otherwise); every pointer and
 * stride a multiple of 16 bytes.  One launch per layer instead of two appends, a mask and a dense SDPA call.
 */
int glb_slab_attention(const void *q, int64_t q_stride_row, int64_t q_stride_head, const void *k_new, int64_t k_stride_row,
                       int64_t k_stride_head, const void *v_new, int64_t v_stride_row, int64_t v_stride_head, void *k_slab,
                       void *v_slab, const int32_t *pos,
                       int64_t n_rows, int64_t heads, int64_t kv_heads, int64_t cap, int64_t head_dim, float scale,
                       int32_t dtype, void *out, void *hip_stream);

/*
 * Masked attention of the padded batches of short contexts the path feeds the transformer (hf.py:232-281): for every row u,
 * query head h and query position t, softmax over the keys s it may see of q[u,h,t] . k[u,h/G,s] * scale, times v.  q / k / v
 * by element strides {row, head, position} with unit inner stride (the projections' outputs, cached prefixes in front of
 * the keys); mask: bool / uint8 [n_rows, q_len, k_len] by strides (row, query), nonzero = attend - the 4-D mask
 * transformers builds from the padding mask - or null: causal, key s <= t + k_len - q_len.  out [n_rows, q_len, heads,
 * head_dim] contiguous.  float32 accumulation; head_dim 16 / 32 / 64 / 128; pointers and strides multiples of 16 bytes.
 * One wave per (row, head, query): meant for q_len * k_len of a few hundred (a dozen tokens per context), where the
 * library SDPA kernels spend a fifth of a 24 ms step; long sequences stay with them.
 */
int glb_short_attention(const void *q, const int64_t q_strides[3], const void *k, const int64_t k_strides[3], const void *v,
                        const int64_t v_strides[3], const uint8_t *mask, int64_t mask_stride_row, int64_t mask_stride_query,
                        int64_t n_rows, int64_t heads, int64_t kv_heads, int64_t q_len, int64_t k_len, int64_t head_dim,
                        float scale, int32_t dtype, void *out, void *hip_stream);

/*
 * Systematic resampling of the whole population from the all-gathered log-weights (the step the all-gather of
 * README.md:108-110's weights exists for; the reference itself stops at the normalised weights).  Exact integer
 * comb over the fixed-point weights of glb_normalize_weights with one Philox draw keyed by (seed, offset):
 * every rank that holds the same gathered vector computes the same ancestors.
 *   out_ancestors [n] int32 (non-decreasing); out_stats [1] optional: logsumexp of the weights
 *   workspace: glb_resample_workspace(n) bytes, 8-byte aligned.  n <= 262144.
 */
size_t glb_resample_workspace(int64_t n);
int glb_resample_systematic(const float *log_weights, int64_t n, uint64_t seed, uint64_t offset,
                            int32_t *out_ancestors, float *out_stats, void *workspace, size_t workspace_bytes,
                            void *hip_stream);

/*
 * Token -> byte trie masses.  Replaces TokenCharacterTrie.weight_sum / weight_max and their batch_
 * forms (trie/base.py:147-213 with the numba loops at :346-393; trie/parallel.py:92-145): for every node of the
 * trie over the vocabulary's byte strings, the sum / maximum of the weights of the tokens below it, for a batch of
 * weight rows at once.  The trie arrives flattened (built once per vocabulary on the host):
 *   leaf_node   [vocab]        device: node id of token k's leaf
 *   level_start [n_levels + 1] HOST array (it sizes the launches): level_nodes[level_start[d] .. level_start[d+1]) are
 *   level_nodes                the internal nodes d levels above the deepest ones (device); a node's children always sit
 *                              in an earlier level or are leaves
 *   child_ptr   [n_nodes + 1], child_idx: device CSR of every node's children, ascending (the reference's `jump`)
 * out[r, node] = the value as float32; a node's children are added in ascending order in double and the result is
 * rounded to float32 per node (the reference keeps doubles: agreement to ~1e-7 relative per level).
 * from_logprobs != 0: weights = exp(row).  GLB_TRIE_MAX floors internal nodes at 0 like the reference (:385).
 * One launch for the leaves + one per tree level, all rows each.  Batches of 32 rows or more run on node-major
 * values (coalesced levels) if the caller lends glb_trie_workspace(n_rows, n_nodes) bytes of device scratch (16-byte
 * aligned; workspace may be null: the row-major kernels then serve any batch, a few times slower for large ones).
 */
enum { GLB_TRIE_SUM = 0, GLB_TRIE_MAX = 1 };
size_t glb_trie_workspace(int64_t n_rows, int64_t n_nodes);
int glb_trie_reduce(const float *weights, int64_t ld, int64_t n_rows, int64_t vocab, int64_t n_nodes, int64_t n_levels,
                    const int32_t *leaf_node, const int32_t *level_start_host, const int32_t *level_nodes,
                    const int32_t *child_ptr, const int32_t *child_idx, int32_t op, int32_t from_logprobs, float *out,
                    int64_t out_ld, void *workspace, size_t workspace_bytes, void *hip_stream);

/*
 * The same propagation with everything the pipeline around it wants (SURVEY.md §8 f2 "fuses with the logprob kernel"):
 *   - weights of any element type; with from_logprobs the leaf weight is exp(x * logit_scale - lse[r]): handing over the
 *     LOGITS and the lse the fused step returned (out_lse) gives the masses of softmax(logits) without a [B, V]
 *     log-probability matrix ever being written or read (lse nullable: then x are log-probabilities / weights);
 *   - three outputs, any subset: `out` row-major [n_rows, n_nodes]; `out_sel` [n_rows, n_sel], only the nodes a caller
 *     asks for (the children of the nodes its particles stand on); keep_node_major: nothing is transposed back, the
 *     values stay in `workspace` as float32 [n_nodes, pitch], pitch = n_rows rounded up to 64 (the 795 MB transposed
 *     copy of a 1024 x 194k-node batch is more than the propagation itself).
 * workspace: glb_trie_workspace_ex(n_rows, n_nodes) bytes, 16-byte aligned; without it only `out` is served (row-major
 * kernels).
 * The tree is whatever the arrays say: a caller may hand over its trie with the one-child nodes folded away (such a node
 * has its child's value, bit for bit; a byte trie of a BPE vocabulary is mostly such chains: 194 k nodes, 66 k of them
 * leaves or branching) - n_nodes then counts the remaining "slots", leaf_node / level_nodes / child_idx name slots, and
 * sel_nodes maps the nodes the caller wants (all of them, for the reference's [n_rows, n_nodes] result) to their slots.
 * The propagation then moves a third of the bytes (genlm_backend_amd.trie.TokenByteTrie.compact does this).
 */
typedef struct glb_trie_args {
  uint32_t struct_size;  /* sizeof(glb_trie_args) - ABI guard */
  const void *weights;   /* [n_rows, ld] device */
  int32_t dtype;         /* GLB_F32 / GLB_BF16 / GLB_F16 */
  int64_t ld, n_rows, vocab;
  const float *lse;      /* [n_rows] device, nullable */
  float logit_scale;     /* from_logprobs: x * logit_scale (1/temperature) */
  int32_t from_logprobs;
  int32_t op;            /* GLB_TRIE_SUM / GLB_TRIE_MAX */
  int64_t n_nodes, n_levels;
  const int32_t *leaf_node, *level_start_host, *level_nodes, *child_ptr, *child_idx; /* as glb_trie_reduce */
  float *out;            /* nullable */
  int64_t out_ld;
  const int32_t *sel_nodes; /* [n_sel] device */
  int64_t n_sel;
  float *out_sel;        /* nullable */
  int64_t out_sel_ld;
  int32_t keep_node_major;
  void *workspace;
  size_t workspace_bytes;
} glb_trie_args;
size_t glb_trie_workspace_ex(int64_t n_rows, int64_t n_nodes);
int glb_trie_masses(const glb_trie_args *args, void *hip_stream);

/*
 * The same masses with ONE ROW's values of a part of the trie resident in LDS (round 4; glb_trie.hip).  Replaces the
 * batch paths of trie/base.py:196-216 and trie/parallel.py:92-145 for row-major results: the weights are read once
 * and the output written once - no node-major scratch, no transposes.  `plan` is the folded trie cut into parts of at
 * most 160 KB / 6 slots (a slot takes a float32 value and a 16-bit child pointer in LDS;
 * genlm_backend_amd.trie.TokenByteTrie.plan builds it; every array lives on the device):
 *   desc [n_parts + 1][16] int32, per part: slot_base, n_local, n_roots, n_depths, (unused), cptr_off, leaf_off,
 *        n_leaves, cut_base, node_off, n_nodes, inode_off, n_inodes, idepth_off, run_off, n_runs; part n_parts is the
 *        top when n_top > 0
 *   local slots of a part are numbered breadth first over its subtrees: the children of local slot s are
 *        cptr16[cptr_off + s] .. cptr16[cptr_off + s + 1] (ascending child order; cptr_off even)
 *   inode16 [inode_off ..+ n_inodes] (inode_off even): the part's internal local slots depth by depth; depth k is
 *        idepth[idepth_off + k] .. idepth[idepth_off + k + 1] of them, its children all sit in depth k + 1
 *   leaf_src / leaf_local [leaf_off ..+ n_leaves]: the part's tokens in ascending order (top: indices of cut roots) and
 *        the local slots their weights go to
 *   the trie nodes whose value a local slot of the part holds are n_runs runs of consecutive node ids (a subtree is an
 *        interval of the post-order numbering): run_tab [run_off ..+ n_runs][2] = (first node, count), and
 *        pn_local16 [node_off ..+ n_nodes]: the local slots of those nodes, run after run
 *   top_local [n_top]: local slot of top slot top_base + i;  slot_of [n_nodes]: node -> slot (parts first, top last)
 *   lds_bytes: the largest part's 4 n_local + 2 (n_local + 1, rounded up to even) + 2 (n_inodes, rounded up to even)
 *        + 128 (its depth table: n_depths <= 30)
 * Same arithmetic per node as glb_trie_reduce (children in ascending order, accumulated in double, stored as float32),
 * so the results are bit-equal to glb_trie_masses on the same folded trie.
 * Outputs (any subset): out_slots [n_rows, n_slots] in the plan's slot numbering; out_nodes [n_rows, n_nodes];
 * out_sel [n_rows, n_sel] = the nodes sel_nodes[0 .. n_sel), n_sel <= 2^20 - or, with sel_row_stride, every row's own
 * nodes.  workspace: glb_trie_rows_workspace bytes (the values of the parts' subtree roots, read by the top's launch; the
 * slots of the selected nodes; a row's needed parts).
 */
typedef struct glb_trie_plan {
  uint32_t struct_size;  /* sizeof(glb_trie_plan) - ABI guard */
  int32_t n_parts, n_top, n_cut, n_slots, max_local, top_base, lds_bytes;
  int64_t n_nodes;
  const int32_t *desc, *idepth, *leaf_src, *leaf_local, *run_tab, *top_local, *slot_of;
  const uint16_t *cptr16, *inode16, *pn_local16;
  /* ABI 8 - a SWEEP plan (tok_local16 non-null; TokenByteTrie.plan(sweep=True)): a (row, part) workgroup reads the whole
   * row front to back instead of gathering the part's tokens; a persistent workgroup keeps a part's values in LDS
   * (lds_bytes >= 4 * max_local + 128: parts of up to 40 000 slots), the tokens' slots and the part's internal nodes in
   * registers, and has the head of the next row on its way while it reduces and writes the current one:
   *   tok_local16 [n_parts][(vocab + 7) & ~7]: token -> its local slot in that part; another part's token -> a word of the
   *            32-word slack behind the part's values, n_local + (token / 8) % 32 (the kernel stores every token's weight);
   *   inode64: per part at desc[D_INODE_OFF], the part's internal nodes as inode16 lists them (depth by depth):
   *            local slot | first child << 16 | number of children << 32;
   *   lds_top_bytes: the LDS of the launch over the top (its values and 16-bit tables, as for a gathered part);
   *   vocab: the vocabulary tok_local16 was made for (glb_trie_rows_args.vocab must equal it).
   * All null / 0: the gathered plan of rounds 4-5. */
  const uint16_t *tok_local16;
  const uint64_t *inode64;
  int32_t lds_top_bytes, vocab;
} glb_trie_plan;
typedef struct glb_trie_rows_args {
  uint32_t struct_size;  /* sizeof(glb_trie_rows_args) - ABI guard */
  const void *weights;   /* [n_rows, ld] device */
  int32_t dtype;         /* GLB_F32 / GLB_BF16 / GLB_F16 */
  int64_t ld, n_rows, vocab;
  const float *lse;      /* [n_rows] device, nullable (from_logprobs: exp(x * logit_scale - lse[r])) */
  float logit_scale;
  int32_t from_logprobs;
  int32_t op;            /* GLB_TRIE_SUM / GLB_TRIE_MAX */
  float *out_slots;      /* nullable */
  int64_t out_slots_ld;
  float *out_nodes;      /* nullable */
  int64_t out_nodes_ld;
  const int32_t *sel_nodes; /* [n_sel] device */
  int64_t n_sel;
  float *out_sel;        /* nullable */
  int64_t out_sel_ld;
  void *workspace;
  size_t workspace_bytes;
  int64_t sel_row_stride; /* 0: sel_nodes [n_sel] is one selection for every row.  > 0: a selection PER ROW - sel_nodes
                             [n_rows, sel_row_stride], row r asks for its first n_sel entries (a negative entry: nothing,
                             its output is 0; n_rows * n_sel <= 2^20) - e.g. every particle's current node's children, what a
                             byte-level sampler reads after trie/base.py:147-213.  Only the parts of the trie that hold a row's
                             nodes are read and reduced for that row (every part when a node sits above the cut). */
} glb_trie_rows_args;
size_t glb_trie_rows_workspace(int64_t n_rows, const glb_trie_plan *plan);
int glb_trie_rows(const glb_trie_rows_args *args, const glb_trie_plan *plan, void *hip_stream);

/*
 * torch's CPU generator for GLB_RNG_NOISE (parity mode).  The reference draws every token with torch.multinomial on the
 * CPU (README.md:87, base.py:136-141): V float32 Exp(1) variates per particle from ONE serial MT19937 stream - each
 * variate is one random64() = two MT words, u = (r & (2^53 - 1)) 2^-53, E = (float)(-log1p(-u)) - particles in
 * resolution order (hf.py:285-288).
 *
 * Host, serial (validation, small cases): glb_mt19937_seed + glb_mt19937_exponential_f32 fill out[0..n) with what
 * torch.empty(n).exponential_(1, generator) produces for a generator seeded alike.  Host pointers only.
 */
typedef struct glb_mt19937 {
  uint32_t mt[624];
  int32_t idx;
} glb_mt19937;
void glb_mt19937_seed(glb_mt19937 *st, uint64_t seed);
int glb_mt19937_exponential_f32(glb_mt19937 *st, float *out, int64_t n);

/*
 * The same stream on the DEVICE, entered at every particle's row at once (csrc/glb_mt.hip).  The stream's position is a
 * WINDOW: the 624 untempered words x[o .. o+623] that precede the next output (after seeding: the seeded array itself,
 * glb_mt19937_window).  A window `J` words ahead is g_J(T) applied to a window, g_J(t) = t^J mod the generator's minimal
 * polynomial: glb_mt19937_jump_polys computes, on the host, once per stride (stride_words = 2 V: one particle's row),
 *     polys[r]           = g_{r * stride}              r = 0 .. n_small - 1
 *     polys[n_small + m] = g_{m * n_small * stride}    m = 0 .. n_big - 1
 * (GLB_MT_POLY_WORDS 64-bit words each, coefficient i = bit i % 64 of word i / 64; about 2 ms per polynomial), enough
 * for n_small * n_big - 1 rows per call.  glb_mt19937_jump_host applies one polynomial to a window on the host (tests).
 * Only the top bit of a window's word 0 is state; its other 31 bits are unspecified after a jump.
 */
#define GLB_MT_POLY_WORDS 312
int glb_mt19937_window(uint64_t seed, uint32_t *out_window /* [624] host */);
int glb_mt19937_jump_polys(int64_t stride_words, int32_t n_small, int32_t n_big, uint64_t *out_polys /* host */);
int glb_mt19937_jump_host(const uint32_t *window_in, const uint64_t *poly, uint32_t *window_out);
/*
 * out[i, 0..V) = the V exponentials of stream row row_slot[i] (row k = words [2 V k, 2 V (k + 1)) after the position
 * `window`); row_slot[i] < 0: a row of ones (a particle that draws nothing this step); row_slot null: identity.  Then
 * window_out (nullable; may be `window` itself) = the position after *n_draw rows (n_draw: device scalar <= max_draw_rows,
 * null: max_draw_rows) - the stream moves on by what was consumed, decided on the device.  Three launches, no host
 * synchronisation.  All pointers device pointers.  When the output rows take at most half of the stream's rows (a rank of a
 * sharded population that enters ONE global stream: n_out_rows of max_draw_rows), only the windows some output row - or
 * window_out - reads are made (a word per row in the workspace says which).
 */
typedef struct glb_mt_rows_args {
  uint32_t struct_size;
  const uint32_t *window;   /* [624] */
  uint32_t *window_out;     /* [624], nullable */
  const uint64_t *polys;    /* [(n_small + n_big), GLB_MT_POLY_WORDS] as glb_mt19937_jump_polys made them for stride 2 * vocab */
  int32_t n_small, n_big;
  int64_t vocab;            /* V */
  int64_t max_draw_rows;    /* host: upper bound of the stream rows this call consumes; < n_small * n_big */
  const int32_t *n_draw;    /* device scalar, nullable */
  int64_t n_out_rows;
  const int32_t *row_slot;  /* [n_out_rows], nullable */
  float *out;               /* [n_out_rows, out_ld] */
  int64_t out_ld;           /* >= vocab */
  void *workspace;          /* >= glb_mt19937_rows_workspace(max_draw_rows, n_small) bytes */
  size_t workspace_bytes;
  /* Two-phase use (ABI 8), for a caller that generates the NEXT step's rows ahead of time on another stream - they do not
   * depend on the step's data, only on where the stream stands -: call 1 (ahead): row_slot null, n_out_rows = the rows
   * that may be needed, out = a buffer G, window_out null.  Call 2 (when the step knows its row_slot and n_draw), same
   * window / polys / max_draw_rows / workspace: reuse_windows = 1 (the workspace still holds every row's window: the two
   * jump launches are skipped), rows_from = G (out rows are COPIED from G[row_slot[i]], ones for a negative slot, instead of
   * generated), window_out = the position after *n_draw rows.  Both 0 / null: one call does everything. */
  int32_t reuse_windows;
  const float *rows_from;   /* [>= max_draw_rows, rows_from_ld] */
  int64_t rows_from_ld;
} glb_mt_rows_args;
size_t glb_mt19937_rows_workspace(int64_t max_draw_rows, int32_t n_small);
int glb_mt19937_exponential_rows(const glb_mt_rows_args *args, void *hip_stream);

/*
 * The path's one collective for a caller WITHOUT PyTorch (SURVEY.md §8(b) sketched glb_allgather_f32(comm, ...)): once the
 * particles are split over GPUs, README.md:108-110's normalisation needs every shard's log-weights on every rank - an
 * all-gather of n floats per rank over RCCL / xGMI (4 KiB at 1024 particles: latency-bound).  Rank 0 makes an id
 * (glb_comm_unique_id, GLB_COMM_ID_BYTES bytes) and hands it to the other ranks by whatever channel the integrator has;
 * every rank calls glb_comm_init with its device current; recv holds world * n floats, rank-major.  RCCL is taken from the
 * process at run time (dlopen); GLB_EUNSUPPORTED when there is none.  The Python host uses torch.distributed ("nccl" is
 * the same RCCL) and never calls these.
 */
#define GLB_COMM_ID_BYTES 128
int glb_comm_unique_id(void *out_id);
int glb_comm_init(const void *id, int32_t rank, int32_t world, void **out_comm);
int glb_allgather_f32(void *comm, const float *send, int64_t n, float *recv, void *hip_stream);
int glb_comm_destroy(void *comm);

/* Philox4x32-10 block function, exposed so hosts can reproduce the device draws. */
void glb_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* GLB_H */
