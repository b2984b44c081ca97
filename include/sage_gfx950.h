/*
 * sage_gfx950.h -- C ABI of libsage_gfx950.so: SageAttention's quantized fused-attention hot
 * path for AMD MI355X (gfx950 / CDNA4).
 *
 * This is the drop-in boundary.  Each entry point replaces one (or a family) of the reference's
 * native ops -- the pybind11 functions behind `sageattention._fused` and
 * `sageattention._qattn_sm80/_sm89/_sm90` -- cited per function as file:line relative to the
 * thu-ml/SageAttention tree.  The contract mirrors theirs:
 *   - the CALLER owns every buffer (outputs are pre-allocated, the library only writes into
 *     them; reference: `mutates_args=("output",)`, sageattention/sm80_compile.py:5);
 *   - the library keeps no state, allocates nothing, and is re-entrant; work is enqueued on the
 *     HIP stream passed as `stream` (a hipStream_t; NULL = the null stream) and the call returns
 *     without synchronising;
 *   - integer encodings follow the reference's op boundary: tensor dtype 0 = fp16, 1 = bf16;
 *     `qk_quant_gran` 2 = per_warp, 3 = per_thread (core.py:556-559), plus 1 = per_block
 *     (the Triton path's granularity);
 *   - errors: every function returns 0 on success or a negative SAGE_E* code, and
 *     sage_last_error() returns a thread-local human-readable message (the reference raises
 *     through TORCH_CHECK / std::invalid_argument, csrc/utils.cuh:20-37).
 * All strides are in ELEMENTS.  Pointers are device pointers unless stated otherwise and must
 * be 16-byte aligned; row strides must be multiples of 16 elements (the innermost/head_dim
 * stride is 1, as the reference asserts, core.py:274).  head_dim is 64 or 128 (the Python layer
 * pads other sizes exactly as core.py:260-271 does).
 */
#ifndef SAGE_GFX950_H
#define SAGE_GFX950_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAGE_ABI_VERSION 21

#if defined(__GNUC__)
#define SAGE_API __attribute__((visibility("default")))
#else
#define SAGE_API
#endif

#define SAGE_OK 0
#define SAGE_EINVAL (-1)      /* bad argument (shape, alignment, unsupported combination) */
#define SAGE_ELAUNCH (-2)     /* HIP launch / runtime error */

#define SAGE_DTYPE_F16 0
#define SAGE_DTYPE_BF16 1

/* qk_quant_gran at the op boundary (reference core.py:556-559; 1 is the Triton path's) */
#define SAGE_GRAN_PER_BLOCK 1
#define SAGE_GRAN_PER_WARP 2
#define SAGE_GRAN_PER_THREAD 3
#define SAGE_GRAN_KBLK128 0x100      /* OR-ed into qk_quant_gran: k scale groups span 128 keys (sm90 kernels, core.py:964-970) */

/* rounding / epsilon convention of the INT8 quantiser */
#define SAGE_QSTYLE_TRITON 0         /* quant_per_block.py:39-47: x/scale, round half away, no eps */
#define SAGE_QSTYLE_CUDA 1           /* fused.cu:147-186: amax floor 1e-7, x*(127/amax), RNE sat   */
#define SAGE_QSTYLE_TRITON_THREAD 2  /* quant_per_thread.py:41-44: scale = amax/127 + 1e-7         */

/* PV accumulation */
#define SAGE_PV_ACCUM_SINGLE 0       /* accumulate every tile straight into the FP32 output          */
#define SAGE_PV_ACCUM_TWO_LEVEL 1    /* per-64-key tile product from zero, then added to FP32 output */
#define SAGE_PV_ACCUM_TRITON 2       /* FP16 PV only: the reference's Triton kernel form (attn_qk_int8_per_block.py:53-62): the tile
                                        product is folded into the FP32 output and the softmax denominator sums the UN-rounded
                                        probabilities.  0 / 1 on an FP16-PV entry point select its CUDA kernel form
                                        (qk_int_sv_f16_cuda_sm80.cu:303-320): FP32 accumulation (gfx950 MFMAs have no FP16
                                        accumulator; both values run alike) and the denominator sums the fp16-ROUNDED
                                        probabilities, as the reference's tensor-core row sum does. */

SAGE_API int sage_abi_version(void);
SAGE_API const char *sage_last_error(void);

/* Work order of causal dense launches of the 128-row kernels (which (head, query block) item workgroup blockIdx takes; results do
 * not depend on it).  -1 (default): heads in groups sized by the grid, longest query blocks of a group first, single-round grids
 * folded so that a CU's two workgroups are a long and a short block; 0: head-major, longest block of each head first (rounds 1-2);
 * n > 0: groups of n heads.  The reference launches blockIdx.x = query block in ascending order (qk_int_sv_f8_cuda_sm89.cuh:720-738)
 * and leaves the order to the hardware.  DEBUG / A-B SWITCH, not part of the data path's contract: one process-wide integer, initialised from the
 * environment variable SAGE_ORDER_GROUP, read by every later launch; NOT thread-safe against concurrent launches (set it before the first call
 * or not at all).  Apart from this switch and sage_debug_prepass_fail (a test hook) the library keeps no state between calls. */
SAGE_API int sage_work_order(void);
SAGE_API void sage_set_work_order(int group);
/* Host-side views of that order (csrc/sage_work_order.h, the code the kernels and launchers run; no GPU needed): the plan of a causal
 * launch over nheads = B * Hq heads of nqblk 128-row query blocks -> grid size, *group / *fold / *left; and the (head, rank of the query
 * block: 0 = the longest) that workgroup `bid` of a grid of `nwg` takes (returns 1, or 0 if that workgroup has no item, -1 on bad arguments). */
SAGE_API int sage_debug_work_order_plan(int nheads, int nqblk, int64_t kv_len, int head_dim, int pv_fp8, int forced, int *group, int *fold, int *left);
SAGE_API int sage_debug_work_item(int bid, int nwg, int nheads, int nqblk, int group, int fold, int left, int *head, int *qrank);

/* Size in bytes of the tiled V^T image for `n_kv_tiles_total` 64-token tiles (all batches, heads). */
SAGE_API int64_t sage_v_image_bytes(int head_dim, int fp8, int64_t n_kv_tiles_total);

/*
 * INT8 quantisation of Q or K (dense [B,H,L,D] with arbitrary b/h/l strides).
 * Replaces: quant_per_block_int8_cuda, quant_per_block_int8_fuse_sub_mean_cuda,
 *           quant_per_warp_int8_cuda            (csrc/fused/fused.h:19-55, pybind.cpp:24-28)
 *           and the Triton quantisers quant_per_block.py:49-101, quant_per_thread.py:154-203.
 *  x        fp16/bf16 input
 *  mean     nullable [B,H,D] K-smoothing mean, same dtype (strides mean_sb, mean_sh)
 *  out      int8, strides o_sb/o_sh/o_sl
 *  scale    fp32 [B,H,nscale]; nscale = ceil(L/blk) * slots, slots = 1 (per_block),
 *           blk/warp (per_warp), blk/warp*8 (per_thread, is_key=0), blk/warp*4 (per_thread, is_key=1)
 *  blk      rows per block: 128 (Q) or 64 (K);  warp: sub-block rows (32/16 for Q, 64 for K)
 *  pre_scale multiplied into x before quantising (sm_scale*log2e for the Triton-path Q, else 1)
 */
SAGE_API int sage_quant_qk_int8(const void *x, const void *mean, int8_t *out, float *scale,
                       int B, int H, int L, int D,
                       int64_t x_sb, int64_t x_sh, int64_t x_sl,
                       int64_t o_sb, int64_t o_sh, int64_t o_sl,
                       int64_t mean_sb, int64_t mean_sh,
                       int blk, int warp, int gran, int is_key, int style,
                       float pre_scale, int dtype, void *stream);

/* Every index array of a packed-batch (varlen) call from ONE small launch, so that the call never synchronises with the host:
 *   cu_q_scale / cu_k_scale  exclusive prefix sums of ceil(Lq_i / blkq), ceil(Lk_i / blkk) (nseq + 1 entries; cu_q_scale nullable)
 *   seq_order   (nullable) the sequences by descending query length: the unit order of an attention launch WITHOUT a work list
 *   work_items  (nullable) the work list of the attention launch: (sequence, 128-row query block) int32 pairs for every query block that
 *               exists, sorted by descending weight = 64-key tiles the block visits under `is_causal` (ties: sequence index, then the
 *               later block first); the caller allocates 2 * (ceil(sum Lq / 128) + nseq) ints, a host-known bound of the count
 *   slab_first / slab_seq  (nullable, together) the 512-token slabs of sage_prepass_kv_varlen: prefix sums of the slab counts of the nseq
 *               sequences and of two gap segments -- rows cu_k[nseq] .. total_k and rows 0 .. cu_k[0] of the packed tensors, which belong to no
 *               sequence but which `k.mean(dim=0)` (core.py:432-434) still averages over: statistics only -- (nseq + 3 ints), and slab ->
 *               segment (caller allocates ceil(total_k / 512) + nseq + 2 ints; total_k = rows of the packed k / v)
 *   hdr         (needed by work_items / slab_seq) 8 ints: number of work items, then the launch plan over them as for a dense causal
 *               launch of Hq heads (csrc/sage_work_order.h: heads per group -- whole GQA groups --, fold, left-over heads), the number
 *               of slabs, max Lk, sum Lk, 0
 * Results of the attention launch do not depend on either order.  nseq <= sage_varlen_plan_max_seqs(); work_items needs blkq = 128, blkk = 64.
 * Replaces: the torch prefix sums of quant_per_block_varlen.py:68-73 and the `.item()` synchronisations of :75-76; the reference launches
 * ceil(max_seqlen_q / 128) blocks for EVERY sequence and lets the ones past a sequence's end exit (attn_qk_int8_block_varlen.py:98-121).
 * work_items_cap / slab_seq_cap: the number of (sequence, block) pairs work_items holds and of entries slab_seq holds.  The counts the kernel
 * derives from cu_seqlens on the device are clamped to them (hdr reports the clamped counts): a cu_seqlens that is inconsistent with the
 * row counts the buffers were sized from drops work, it never makes this launch or a launch that reads hdr write or read out of bounds. */
SAGE_API int sage_varlen_plan_max_seqs(void);
SAGE_API int sage_varlen_plan(const int32_t *cu_seqlens_q, const int32_t *cu_seqlens_k, int nseq, int total_k, int blkq, int blkk,
                              int is_causal, int Hq, int Hkv, int head_dim, int pv_fp8,
                              int32_t *cu_q_scale, int32_t *cu_k_scale, int32_t *seq_order,
                              int32_t *work_items, int work_items_cap, int32_t *slab_first, int32_t *slab_seq, int slab_seq_cap,
                              int32_t *hdr, void *stream);
/* Host-side view of that work list (the same functions of csrc/sage_work_order.h, run on the host; no GPU needed): lq / lk are HOST arrays of
 * the nseq sequence lengths; items_out receives (sequence, query block) pairs, hdr_out {count, group, fold, left}; returns the grid size of
 * the attention launch or a negative status. */
SAGE_API int sage_debug_varlen_items(const int32_t *lq, const int32_t *lk, int nseq, int is_causal, int Hq, int Hkv, int head_dim, int pv_fp8,
                                     int32_t *items_out, int items_cap, int32_t *hdr_out);

/*
 * Same for packed variable-length batches x[sum L, H, D] (per-block only).
 * Replaces: quant_per_block_varlen.py:60-104.  cu_seqlens [nseq+1] and cu_scale [nseq+1]
 * (prefix sums of ceil(L_i/blk)) are int32 device arrays; scale is [cu_scale[nseq], H].
 * `mean` (nullable) is [H, D] shared by all sequences (core.py:432-434 subtracts it in torch).
 */
SAGE_API int sage_quant_qk_int8_varlen(const void *x, const void *mean, int8_t *out, float *scale,
                              const int32_t *cu_seqlens, const int32_t *cu_scale,
                              int nseq, int max_seqlen, int H, int D,
                              int64_t x_sl, int64_t x_sh, int64_t o_sl, int64_t o_sh, int64_t mean_sh,
                              int blk, float pre_scale, int dtype, void *stream);

/*
 * Workspace size, in floats, of the per-channel statistics pass used by sage_channel_mean and
 * sage_prep_v_fp8 for a [B,H,L,D] tensor: partial (max,min,sum) per 512-token slab + the final block.
 */
SAGE_API int64_t sage_stats_ws_floats(int B, int H, int L, int D);

/*
 * Mean over the sequence of x[B,H,L,D] (strides), written in the input dtype to mean_out[B,H,D]
 * (fp32 accumulation, one rounding; deterministic order).  Replaces the torch reduction
 * `km = k.mean(dim=seq)` (core.py:280,584,773; `k.mean(dim=0)` for varlen with B=1, H stride D)
 * and `vm = v.mean(...)` of sub_mean (quant.py:216).  ws: sage_stats_ws_floats(B,H,L,D) floats.
 */
SAGE_API int sage_channel_mean(const void *x, void *mean_out, float *ws, int B, int H, int L, int D,
                               int64_t x_sb, int64_t x_sh, int64_t x_sl, int dtype, void *stream);

/*
 * V pre-pass, FP8: per-channel scale + e4m3 + transpose into the gfx950 PV tile image.
 * Replaces: transpose_pad_permute_cuda + scale_fuse_quant_cuda / mean_scale_fuse_quant_cuda
 *           (csrc/fused/fused.h:57-75, quant.py:224-293).  v is [B,H,L,D] (strides), v_image receives
 *           sage_v_image_bytes(D, 1, B*H*ceil(L/64)) bytes, v_scale [B,H,D] fp32.
 *           v_mean: NULL, or [B,H,D] fp32 out => smooth_v: the per-channel mean (sum / ceil16(L), as
 *           fused.cu:335,381 computes it) is subtracted before quantising and returned for the
 *           attention epilogue.  ws: sage_stats_ws_floats(B,H,L,D) floats of scratch.
 *           scale_max = 448 (e4m3 max).
 */
SAGE_API int sage_prep_v_fp8(const void *v, void *v_image, float *v_scale, float *v_mean, float *ws,
                    int B, int H, int L, int D,
                    int64_t v_sb, int64_t v_sh, int64_t v_sl,
                    float scale_max, int dtype, void *stream);

/*
 * The whole K / V pre-pass of the FP8-PV entry points as ONE launch that reads K and V once (2 B/elt in, 1 B/elt out):
 *   K: km = k.mean(dim=seq) (core.py:280) -> k_mean [B,H,D] (input dtype), INT8 (k - km) + group scales as
 *      sage_quant_qk_int8 writes them for is_key = 1, blk = warp = k_blk, qk_quant_gran per-block or per-thread; k_style: the CUDA
 *      convention or the per-thread Triton one (the reference's CUDA entry points), or SAGE_QSTYLE_TRITON with per-block scales (its
 *      Triton-named API, quant_per_block.py:21-46);
 *      k_mean == NULL: no smoothing (smooth_k = False)
 *   V: v_image / v_scale / v_mean exactly as sage_prep_v_fp8 (v_mean != NULL => smooth_v); v_fp16 != 0: the fp16 image of
 *      sage_prep_v_f16 instead (`v.to(float16)`, core.py:297-298,613; no statistics, v_scale / v_mean unused)
 * Either of k / v may be NULL to run one half.  Results are bit-identical to the sage_channel_mean +
 * sage_quant_qk_int8 + sage_prep_v_fp8 sequence (6 launches, 4 B/elt read): same per-slab summation order.
 *   ws    sage_prepass_ws_floats(B,H,L,D) floats of scratch
 *   sync  sage_prepass_sync_words(B,H) uint32, private to the call while it runs, ZERO ON ENTRY.  The kernel returns the counters it
 *         used to zero before it ends (also when a workgroup gave up), so a buffer that its owner zeroed once serves every later call
 *         issued in stream order -- ABI 18 zeroed it with a launch of its own in front of every call (4.8 us and a kernel boundary);
 *         two launches that may run concurrently need two buffers.  Layout: 32 words per (K|V, b, h):
 *         [0] arrivals, [1] departures, [2] give-up flag (sticky: the one word the kernel never clears): set if a workgroup waited
 *         30 ms of wall-clock time (and at least 2^14 polls: time spent context-switched out does not count) for the other slabs of its
 *         head in vain (the co-residency assumption below was violated).  Such a workgroup computes the head's statistics itself --
 *         it re-reads the whole head, slab by slab, through the same summation order -- so the outputs are the same bits as ever; the
 *         launch is slow, not wrong (rounds 2-3 wrote NaN instead).  sage_prepass_failed_heads(sync, B, H, stream) synchronises the
 *         stream and returns how many (K|V, b, h) entries carry the flag (0 = nobody had to).
 *   L     at most sage_prepass_max_seqlen() (65536 on a whole MI355X; 512 x the CU count on a smaller partition): the slabs of a
 *         head wait for each other inside the launch, so all of them must fit on the device at once; longer
 *         sequences take the three-call sequence (SAGE_EINVAL here).  The bound uses the compute units `stream` may use
 *         (hipExtStreamGetCUMask): sage_prepass_max_seqlen_stream(stream) is that bound.  Per head, ((ceil(L/512)*512 - 1) * row
 *         stride + D) * 2 must stay below 2^32.
 *   host_flag  nullable: a device-visible pinned HOST word (hipHostMalloc).  A workgroup that gives up also stores 1 there
 *         (system scope), so the caller can notice a slow launch at its next call without synchronising -- the Python layer
 *         then warns and routes the device's later calls through the three-call sequence.  What the bound above cannot see is
 *         compute units held by OTHER streams' kernels (e.g. RCCL) for longer than the wait.
 */
/* K-smoothing mean of a packed batch x[sum L, H, D] over ALL tokens (core.py:432-434) -> mean_out [H, D], summed over the per-sequence
 * 512-token slabs of sage_varlen_plan (slab_first / slab_seq / hdr) -- the partition, hence the bits, of sage_prepass_kv_varlen.
 * ws: sage_stats_ws_floats(1, H, 512 * nslab_bound, D) floats. */
SAGE_API int sage_channel_mean_varlen(const void *x, void *mean_out, float *ws, const int32_t *cu_seqlens, const int32_t *slab_first,
                                      const int32_t *slab_seq, const int32_t *hdr, int nseq, int total_tokens, int nslab_bound, int H, int D,
                                      int64_t x_sl, int64_t x_sh, int dtype, void *stream);
SAGE_API int64_t sage_prepass_ws_floats(int B, int H, int L, int D);
SAGE_API int64_t sage_prepass_sync_words(int B, int H);
SAGE_API int sage_prepass_max_seqlen(void);
SAGE_API int sage_prepass_max_seqlen_stream(void *stream);
/* One device-visible word of pinned host memory (zeroed) for `host_flag`: *host_ptr for the host's reads, *device_ptr for the kernels.
 * Owned by the caller (sage_host_word_free); the library keeps no handle to it. */
SAGE_API int sage_host_word_alloc(void **host_ptr, void **device_ptr);
SAGE_API int sage_host_word_free(void *host_ptr);
SAGE_API int sage_prepass_failed_heads(const uint32_t *sync, int B, int H, void *stream);
/* test hook: non-zero makes every following sage_prepass_kv launch wait for a slab that does not exist and give up after
 * 2^10 polls, i.e. every workgroup takes the recompute path above (process-wide; reset with 0) */
SAGE_API void sage_debug_prepass_fail(int on);
SAGE_API int sage_prepass_kv(const void *k, const void *v, void *k_mean, int8_t *k_int8, float *k_scale,
                    void *v_image, float *v_scale, float *v_mean, float *ws, uint32_t *sync,
                    int B, int H, int L, int D,
                    int64_t k_sb, int64_t k_sh, int64_t k_sl, int64_t v_sb, int64_t v_sh, int64_t v_sl,
                    int64_t ko_sb, int64_t ko_sh, int64_t ko_sl,
                    int k_blk, int qk_quant_gran, int k_style, float scale_max, int v_fp16, int dtype, uint32_t *host_flag, void *stream);

/* The K / V pre-pass of sageattn_varlen (core.py:431-444) as ONE launch that reads K and V once: packed k / v [sum L, H, D];
 *   K: km = k.mean(dim=0) over ALL packed tokens -> k_mean [H, D] (input dtype; NULL: no smoothing), then per sequence the INT8
 *      (k - km) rows and one scale per 64 keys with the Triton rounding, k_scale [cu_k_scale[nseq], H] -- the bits of
 *      sage_channel_mean (over the same slabs) + sage_quant_qk_int8_varlen;
 *   V: the fp16 tile image [cu_k_scale[nseq], H, D, 64] of sage_prep_v_f16_varlen (v NULL: K half only).
 * A slab is 512 tokens of ONE sequence (so scale blocks and V tiles never straddle slabs); cu_k_scale, slab_first, slab_seq and hdr come
 * from sage_varlen_plan.  nslab_bound = the host-known bound ceil(total_tokens / 512) + nseq + 2 sizes the grid and the workspace
 * (ws: 2 * H * nslab_bound * 3 * D floats = sage_prepass_ws_floats(1, H, 512 * nslab_bound, D); sync: sage_prepass_sync_words(1, H)).
 * With k_mean the slabs of a head -- all sequences -- wait for each other inside the launch: nslab_bound must not exceed 128 nor the compute
 * units `stream` may use (SAGE_EINVAL otherwise: take the three-call sequence).  A workgroup that gives up recomputes, as in sage_prepass_kv. */
SAGE_API int sage_prepass_kv_varlen(const void *k, const void *v, void *k_mean, int8_t *k_int8, float *k_scale, void *v_image,
                                    float *ws, uint32_t *sync, const int32_t *cu_seqlens_k, const int32_t *cu_k_scale,
                                    const int32_t *slab_first, const int32_t *slab_seq, const int32_t *hdr,
                                    int nseq, int total_tokens, int max_seqlen_k, int nslab_bound, int H, int D,
                                    int64_t k_sl, int64_t k_sh, int64_t v_sl, int64_t v_sh, int64_t ko_sl, int64_t ko_sh,
                                    int dtype, uint32_t *host_flag, void *stream);
/* test hook: nwg workgroups of 1024 threads spin for ms milliseconds on `stream` (two of them fill a compute unit's wave slots) */
SAGE_API int sage_debug_spin(int ms, int nwg, void *stream);

/* V pre-pass, FP16: (bf16 -> fp16) + transpose into the tile image.  Replaces `v.to(float16)`
 * (core.py:297-298,613) and, with v_mean != NULL ([B,H,D] fp32 to subtract), sub_mean_cuda
 * (csrc/fused/fused.h, quant.py:182-222).  Dense. */
SAGE_API int sage_prep_v_f16(const void *v, void *v_image, const float *v_mean, int B, int H, int L, int D,
                    int64_t v_sb, int64_t v_sh, int64_t v_sl, int dtype, void *stream);

/* Same for packed v[sum L, H, D]; cu_tiles = prefix sums of ceil(L_i/64); the image holds
 * cu_tiles[nseq]*H tiles ordered [tile, head]. */
SAGE_API int sage_prep_v_f16_varlen(const void *v, void *v_image, const int32_t *cu_seqlens,
                           const int32_t *cu_tiles, int nseq, int max_seqlen, int H, int D,
                           int64_t v_sl, int64_t v_sh, int dtype, void *stream);

/*
 * Launch attributes of the attention entry points below: every sage_attn_* function takes a trailing `const SageLaunchAttr *attr`
 * (NULL = all defaults).  The attributes are ARGUMENTS of the call they travel with -- the library keeps nothing between calls (ABI 18's
 * thread-local sage_attn_launch_ws setter is gone).  The reference has no counterpart: its kernels leave the order of their thread blocks
 * to the hardware (qk_int_sv_f8_cuda_sm89.cuh:720-738) and have one score form.  Results do not depend on launch_ws / flags bit 1.
 *
 *  struct_bytes     sizeof(SageLaunchAttr) as the caller compiled it (fields past it read as zero; 0 is taken as the full struct)
 *  flags            SAGE_ATTR_* below
 *  launch_ws        nullable: sage_attn_launch_ws_bytes() bytes of device memory, 128-byte aligned, ZERO when this launch starts (zeroed by the
 *                   caller in stream order before its first use) and untouched by anyone else until the launch has finished.  A launch that
 *                   uses it returns every word to zero before it ends (ABI 20: the last workgroup to leave re-arms the counters; ABI 19 left
 *                   them dirty and wanted a memset per call), so ONE block per stream, zeroed once, serves every launch on that stream.
 *                   With it a NON-CAUSAL, unmasked launch of at least twelve rounds of workgroups (and the packed route's causal launch)
 *                   runs as a persistent launch: as many workgroups as the device holds at once take the work items as tickets from 32 queues
 *                   (4 per XCD, own XCD first: the L2 locality of the work order; then the fullest other queue: the XCDs of a device run a few
 *                   per cent apart) -- 2.2-2.8 % faster on the CogVideoX shape and on packed batches
 *                   (profiles/r4_run_p_attention_phase_trace.txt).  Every other launch ignores it (masked and split-KV entry points always).
 *                   A block that is NOT zero makes the launch skip work items: the contract is the caller's (after a launch that did not run
 *                   to its end -- a device fault -- zero it again).
 *  launch_ws_bytes  size of that block (checked)
 *  grid_out         nullable HOST pointer: receives the number of workgroups launched (a persistent launch has fewer than work items)
 *  trace, trace_wgs debug: read by -DSAGE_ATTN_TRACE=1 builds only (tools/attn_trace.py); 16 words per logical workgroup
 */
typedef struct SageLaunchAttr {
    uint32_t struct_bytes;
    uint32_t flags;
    void *launch_ws;
    int64_t launch_ws_bytes;
    int32_t *grid_out;
    uint32_t *trace;
    int32_t trace_wgs;
    int32_t reserved;
} SageLaunchAttr;
/* The softmax argument of a score is fma(s, c, -m) with s the INT32 dot product, c the dequantisation scale in the log2 domain and m the
 * running row maximum (attn_utils.cuh:445-449).  The kernels read the accumulator's bit pattern as the float bias + s * 2^-26
 * (bias = 0x3E22F983 as a float, exact), subtract the bias (exact, Sterbenz) and take that FMA: bit for bit the reference's formula, on every
 * input.  That is the DEFAULT of every entry point since ABI 20 (attr == NULL included), for FP8 and FP16 PV.
 * SAGE_ATTR_FP8_FOLDED_SCORES (FP8 PV only, opt-in; ABI 19's default): one FMA per score, fma(bits, c', -(m + bias * c')) with m + bias * c'
 * rounded once per (row, 64-key tile, k scale) -- 3-7 % faster, NOT the reference's arithmetic: an error of up to 0.32 c in the exponent of a
 * (row, tile, k scale) group, c = sm_scale log2(e) q_scale k_scale = the exponent change per INT8 x INT8 score step (1e-4 on randn inputs, 1e-2 with
 * activations of a few tens), which re-rolls e4m3 roundings of P (single outputs move by up to 1.4e-2 * max|o|, rel-RMS 1e-3 ... 2e-3; statistically
 * the same result against fp32 attention) and from c ~ 0.1 on loses whole rows (profiles/r5_run_i_score_scale_range.txt).  A variant for callers who
 * know their magnitudes; the oracle mirrors it (oracle/sage_oracle.c score_mode 1) so that it can be tested, nothing more.
 * SAGE_ATTR_FP8_EXACT_SCORES: what ABI 19 callers set to obtain today's default; still accepted, no effect. */
#define SAGE_ATTR_FP8_EXACT_SCORES 1u
#define SAGE_ATTR_FP8_FOLDED_SCORES 4u
/* tests: take the persistent route from two rounds of workgroups up instead of twelve (needs launch_ws) */
#define SAGE_ATTR_FORCE_PERSISTENT 2u
SAGE_API int64_t sage_attn_launch_ws_bytes(void);

/*
 * Fused attention, INT8 QK^T + FP8 PV.
 * Replaces: qk_int8_sv_f8_accum_f32_fuse_v_scale_attn[_inst_buf],
 *           qk_int8_sv_f8_accum_f16_fuse_v_scale_attn_inst_buf,
 *           qk_int8_sv_f8_accum_f32_fuse_v_scale_fuse_v_mean_attn
 *           (csrc/qattn/attn_cuda_sm89.h:19-104, pybind_sm89.cpp:21-30) and the sm90 ops
 *           (attn_cuda_sm90.h:19-43).
 *  q,k       int8 with strides; q_scale/k_scale as produced by sage_quant_qk_int8 with the same
 *            `qk_quant_gran` (and q_warp = 32 or 16 for per_warp)
 *  v_image   from sage_prep_v_fp8; v_scale [B,Hkv,D]; v_mean nullable [B,Hkv,D]
 *  o         fp16/bf16 (out_dtype) with strides; lse nullable fp32 [B,Hq,Lq] (log2 units + max,
 *            i.e. what the reference kernels return before core.py:823-826 rescales it)
 *  sm_scale_log2  factor applied to dequantised scores: sm_scale*log2e, or 1.0 when it was
 *            folded into Q at quantisation time
 */
SAGE_API int sage_attn_qk_int8_pv_f8(const int8_t *q, const int8_t *k, const void *v_image, void *o, float *lse,
                            const float *q_scale, const float *k_scale,
                            const float *v_scale, const float *v_mean,
                            int B, int Hq, int Hkv, int Lq, int Lk, int D,
                            int64_t q_sb, int64_t q_sh, int64_t q_sl,
                            int64_t k_sb, int64_t k_sh, int64_t k_sl,
                            int64_t o_sb, int64_t o_sh, int64_t o_sl,
                            int is_causal, int qk_quant_gran, int q_warp,
                            float sm_scale_log2, int pv_accum, int out_dtype, void *stream, const SageLaunchAttr *attr);

/* Fused attention, INT8 QK^T + FP16 PV (FP32 accumulate).
 * Replaces: qk_int8_sv_f16_accum_f32_attn, _accum_f16_attn, _accum_f16_attn_inst_buf,
 *           _accum_f16_fuse_v_mean_attn (csrc/qattn/attn_cuda_sm80.h:19-65) and the Triton
 *           `forward` ops attn_qk_int8_per_block.py:130, _causal.py:124. */
SAGE_API int sage_attn_qk_int8_pv_f16(const int8_t *q, const int8_t *k, const void *v_image, void *o, float *lse,
                             const float *q_scale, const float *k_scale, const float *v_mean,
                             int B, int Hq, int Hkv, int Lq, int Lk, int D,
                             int64_t q_sb, int64_t q_sh, int64_t q_sl,
                             int64_t k_sb, int64_t k_sh, int64_t k_sl,
                             int64_t o_sb, int64_t o_sh, int64_t o_sl,
                             int is_causal, int qk_quant_gran, int q_warp,
                             float sm_scale_log2, int pv_accum, int out_dtype, void *stream, const SageLaunchAttr *attr);

/* attn_mask kinds of sage_attn_qk_int8_pv_f16_masked */
#define SAGE_MASK_BOOL 1      /* 1 byte per element, non-zero = attend           */
#define SAGE_MASK_F16 2       /* additive, fp16                                   */
#define SAGE_MASK_BF16 3      /* additive, bf16                                   */

/* FP16-PV attention with an attention mask, non-causal, per-block scales (sm_scale*log2e folded into Q).
 * Replaces: the Triton `forward(..., attn_mask=...)` op (triton/attn_qk_int8_per_block.py:31-51,130-183).
 * mask is addressed as mask[b*m_sb + h*m_sh + q*m_sq + k*m_sk] (element strides; 0 = broadcast, as
 * `attn_mask.expand(...)` produces, core.py:313-322).  Semantics mirror the reference: bool -> 0 / -1e6
 * added to the score and a 128x64 tile whose mask block is all False is skipped; float -> the mask value
 * is added to the (log2-domain) score; out-of-range positions count as False / -1e6. */
SAGE_API int sage_attn_qk_int8_pv_f16_masked(const int8_t *q, const int8_t *k, const void *v_image, void *o, float *lse,
                                    const float *q_scale, const float *k_scale,
                                    const void *mask, int mask_kind,
                                    int64_t m_sb, int64_t m_sh, int64_t m_sq, int64_t m_sk,
                                    int B, int Hq, int Hkv, int Lq, int Lk, int D,
                                    int64_t q_sb, int64_t q_sh, int64_t q_sl,
                                    int64_t k_sb, int64_t k_sh, int64_t k_sl,
                                    int64_t o_sb, int64_t o_sh, int64_t o_sl,
                                    float sm_scale_log2, int out_dtype, void *stream, const SageLaunchAttr *attr);

/* Variable-length fused attention (per-block scales, FP16 PV), packed q/o [sum Lq, Hq, D],
 * k [sum Lk, Hkv, D].  Replaces: attn_qk_int8_block_varlen.py:123, _causal_varlen.py:125.
 * cu_q_scale / cu_k_scale: prefix sums of ceil(Lq_i/128) / ceil(Lk_i/64) (also the V tile prefix).
 * Work order, either of (results do not depend on it):
 *   work_items / work_hdr / items_bound  the work list and header of sage_varlen_plan and the host-known bound of the item count that sized
 *              work_items (ceil(sum Lq / 128) + nseq): every workgroup takes one existing query block, heaviest first, GQA groups per XCD;
 *   seq_order  (without a work list; nullable, device int32[nseq]) a permutation of the sequence indices: (sequence, kv-head) units are
 *              dealt to the XCDs in that order, ceil(max_seqlen_q / 128) blocks per unit, the ones past a sequence's end exit. */
SAGE_API int sage_attn_qk_int8_pv_f16_varlen(const int8_t *q, const int8_t *k, const void *v_image, void *o,
                                    const float *q_scale, const float *k_scale,
                                    const int32_t *cu_seqlens_q, const int32_t *cu_seqlens_k,
                                    const int32_t *cu_q_scale, const int32_t *cu_k_scale, const int32_t *seq_order,
                                    const int32_t *work_items, const int32_t *work_hdr, int items_bound,
                                    int nseq, int max_seqlen_q, int Hq, int Hkv, int D,
                                    int64_t q_sl, int64_t q_sh, int64_t k_sl, int64_t k_sh,
                                    int64_t o_sl, int64_t o_sh,
                                    int is_causal, float sm_scale_log2, int pv_accum, int out_dtype,
                                    void *stream, const SageLaunchAttr *attr);

/* FP8-PV attention (two-level accumulation, "per-thread" granularity) with the Q quantisation fused into the kernel:
 * q is the fp16 / bf16 query tensor itself (element strides); each workgroup quantises its 128 rows in registers with
 * exactly the arithmetic of sage_quant_qk_int8(gran = per_thread, style = SAGE_QSTYLE_TRITON_THREAD, pre_scale = 1), so
 * the result is bit-identical to sage_quant_qk_int8 + sage_attn_qk_int8_pv_f8 while the INT8 copy of Q and its scales
 * never touch HBM (3 B/element of pre-pass traffic and one launch less).  k / k_scale are the per-thread-quantised keys.
 * Replaces: quant_per_thread.py:154-188 (the q half) + qk_int_sv_f8_cuda_sm90.cu / sm89 fuse_v_scale ops called in
 * sequence by core.py:773-821. */
SAGE_API int sage_attn_fused_q_pv_f8(const void *q, const int8_t *k, const void *v_image, void *o, float *lse,
                                     const float *k_scale, const float *v_scale, const float *v_mean,
                                     int B, int Hq, int Hkv, int Lq, int Lk, int D,
                                     int64_t q_sb, int64_t q_sh, int64_t q_sl, int64_t k_sb, int64_t k_sh, int64_t k_sl,
                                     int64_t o_sb, int64_t o_sh, int64_t o_sl,
                                     int is_causal, float sm_scale_log2, int q_dtype, int out_dtype, void *stream, const SageLaunchAttr *attr);

/* FP16-PV attention (FP32 accumulation, "per-thread" granularity) with the same fused Q quantisation: bit-identical to
 * sage_quant_qk_int8 + sage_attn_qk_int8_pv_f16(pv_accum = single).  v_image from sage_prep_v_f16; v_mean nullable [B,Hkv,D].
 * Replaces: quant_per_thread.py:154-188 (the q half) + qk_int8_sv_f16_accum_f32_attn (core.py:598-625). */
SAGE_API int sage_attn_fused_q_pv_f16(const void *q, const int8_t *k, const void *v_image, void *o, float *lse,
                                      const float *k_scale, const float *v_mean,
                                      int B, int Hq, int Hkv, int Lq, int Lk, int D,
                                      int64_t q_sb, int64_t q_sh, int64_t q_sl, int64_t k_sb, int64_t k_sh, int64_t k_sl,
                                      int64_t o_sb, int64_t o_sh, int64_t o_sl,
                                      int is_causal, float sm_scale_log2, int q_dtype, int out_dtype, void *stream, const SageLaunchAttr *attr);

/* The same two fused-Q FP16-PV launches with V READ IN PLACE (ABI 20): `v` is the caller's fp16 value tensor itself -- rows of D halves,
 * last dimension contiguous, element strides v_sb / v_sh / v_sl (multiples of 8), exactly what the reference's FP16-PV ops take
 * ("value fp16, last dim contiguous", qk_int_sv_f16_cuda_sm80.cu:693-704; the Triton forward likewise) -- instead of the gfx950 tile
 * image.  The kernel copies 64-token tiles of rows into LDS by LDS-DMA and forms the PV operand with transposing LDS reads
 * (ds_read_b64_tr_b16).  Bit-identical outputs to the image entry points; for fp16 inputs the V half of the pre-pass (4 of its 7 bytes per
 * element on an FP16-PV call) is not needed at all: core.py:297-298,613's `v.to(torch.float16)` is the identity there.  fp16 q / k / v of
 * one call only (bf16 inputs convert V, i.e. need the image pass); dense, no v_mean, no split.
 *   _fused_q_      : per-thread q / k scale groups, CUDA kernel form (sage_attn_fused_q_pv_f16)
 *   _fused_qblock_ : per-block groups, q_premul folded in, Triton kernel form (sage_attn_fused_qblock_pv_f16) */
SAGE_API int sage_attn_fused_q_pv_f16_vrows(const void *q, const int8_t *k, const void *v, void *o, float *lse, const float *k_scale,
                                            int B, int Hq, int Hkv, int Lq, int Lk, int D,
                                            int64_t q_sb, int64_t q_sh, int64_t q_sl, int64_t k_sb, int64_t k_sh, int64_t k_sl,
                                            int64_t v_sb, int64_t v_sh, int64_t v_sl, int64_t o_sb, int64_t o_sh, int64_t o_sl,
                                            int is_causal, float sm_scale_log2, int out_dtype, void *stream, const SageLaunchAttr *attr);
SAGE_API int sage_attn_fused_qblock_pv_f16_vrows(const void *q, const int8_t *k, const void *v, void *o, float *lse, const float *k_scale,
                                                 int B, int Hq, int Hkv, int Lq, int Lk, int D,
                                                 int64_t q_sb, int64_t q_sh, int64_t q_sl, int64_t k_sb, int64_t k_sh, int64_t k_sl,
                                                 int64_t v_sb, int64_t v_sh, int64_t v_sl, int64_t o_sb, int64_t o_sh, int64_t o_sl,
                                                 int is_causal, float q_premul, int out_dtype, void *stream, const SageLaunchAttr *attr);

/* INT8 q / k with their scale tensors and V READ IN PLACE (ABI 21): argument for argument the reference's native FP16-PV ops --
 *   qk_int8_sv_f16_accum_f32_attn / _f16_attn / _f16_attn_inst_buf / _f16_fuse_v_mean_attn(query i8, key i8, value f16, output, query_scale,
 *   key_scale, [value_mean], tensor_layout, is_causal, qk_quant_gran, sm_scale, return_lse)   csrc/qattn/pybind_sm80.cpp:21-27, attn_cuda_sm80.h:19-65
 * and the kernel-level entry of its Triton path, forward(q, k, v, q_scale, k_scale, ...) (sageattention/triton/attn_qk_int8_per_block.py:130,
 * attn_qk_int8_per_block_causal.py:124: pv_accum = SAGE_PV_ACCUM_TRITON, qk_quant_gran = SAGE_GRAN_PER_BLOCK, sm_scale_log2 = 1) --
 * with `value` the fp16 tensor itself ("value fp16, last dim contiguous", qk_int_sv_f16_cuda_sm80.cu:693-704): rows of D halves, element
 * strides v_sb / v_sh / v_sl (multiples of 8), no tile image and no V pass in front of the call.  Same arguments as
 * sage_attn_qk_int8_pv_f16 otherwise (v_mean: the per-channel mean the fuse_v_mean op adds back, or NULL); outputs and LSE bit-identical to
 * it on the image of the same V.  Dense, unmasked; bf16 value tensors take the image entry (the reference converts them first, core.py:297-298). */
SAGE_API int sage_attn_qk_int8_pv_f16_vrows(const int8_t *q, const int8_t *k, const void *v, void *o, float *lse,
                                            const float *q_scale, const float *k_scale, const float *v_mean,
                                            int B, int Hq, int Hkv, int Lq, int Lk, int D,
                                            int64_t q_sb, int64_t q_sh, int64_t q_sl, int64_t k_sb, int64_t k_sh, int64_t k_sl,
                                            int64_t v_sb, int64_t v_sh, int64_t v_sl, int64_t o_sb, int64_t o_sh, int64_t o_sl,
                                            int is_causal, int qk_quant_gran, int q_warp,
                                            float sm_scale_log2, int pv_accum, int out_dtype, void *stream, const SageLaunchAttr *attr);

/* The Triton-named API's attention (FP16 PV, tile product folded into the FP32 output, per-block k scales) with the PER-BLOCK Q
 * quantisation in the kernel prologue: q (fp16 / bf16) is multiplied by q_premul (= sm_scale * log2 e), one scale per 128 query rows,
 * Triton rounding -- bit-identical to sage_quant_qk_int8 (per_block, pre_scale = q_premul) + sage_attn_qk_int8_pv_f16(pv_accum =
 * SAGE_PV_ACCUM_TRITON, sm_scale_log2 = 1), without the INT8 copy of Q and its scales in HBM.  lse nullable (log2 units).
 * Replaces: quant_per_block.py:21-46 (the q half, core.py:284-286) + attn_qk_int8_per_block*.py forward. */
SAGE_API int sage_attn_fused_qblock_pv_f16(const void *q, const int8_t *k, const void *v_image, void *o, float *lse, const float *k_scale,
                                           int B, int Hq, int Hkv, int Lq, int Lk, int D,
                                           int64_t q_sb, int64_t q_sh, int64_t q_sl, int64_t k_sb, int64_t k_sh, int64_t k_sl,
                                           int64_t o_sb, int64_t o_sh, int64_t o_sl,
                                           int is_causal, float q_premul, int q_dtype, int out_dtype, void *stream, const SageLaunchAttr *attr);
/* The packed / varlen form (sage_attn_qk_int8_pv_f16_varlen's operands without q_scale / cu_q_scale; q in fp16 / bf16).
 * Replaces: quant_per_block_varlen.py:60-104 (the q half, core.py:436-439) + attn_qk_int8_block_varlen.py forward. */
SAGE_API int sage_attn_fused_qblock_pv_f16_varlen(const void *q, const int8_t *k, const void *v_image, void *o, const float *k_scale,
                                                  const int32_t *cu_seqlens_q, const int32_t *cu_seqlens_k, const int32_t *cu_k_scale,
                                                  const int32_t *seq_order, const int32_t *work_items, const int32_t *work_hdr, int items_bound,
                                                  int nseq, int max_seqlen_q, int Hq, int Hkv, int D,
                                                  int64_t q_sl, int64_t q_sh, int64_t k_sl, int64_t k_sh, int64_t o_sl, int64_t o_sh,
                                                  int is_causal, float q_premul, int q_dtype, int out_dtype, void *stream, const SageLaunchAttr *attr);

/* The same kernel over a key range split into kv_split chunks of Lk_chunk keys each (a whole number of 64-key tiles), folded
 * into the kv-head dimension: k / k_scale / v_image / v_scale / v_mean are the operands of the unsplit call viewed as
 * [B, Hkv * kv_split, Lk_chunk, ...] (zero-copy for head-major storage; v_scale / v_mean repeated per chunk), q is the
 * unsplit query tensor [B, Hq, Lq, D] (read in place by every chunk), o_part [B, Hq * kv_split, Lq, D] and lse_part
 * [B, Hq * kv_split, Lq] receive one normalised partial state per chunk in the order sage_merge_split expects
 * (query head index = (hk * kv_split + chunk) * group + g).  is_causal masks key > query row in GLOBAL key coordinates
 * (chunks behind the diagonal produce lse = -inf).  Replaces nothing in the reference (see sage_merge_split). */
SAGE_API int sage_attn_fused_q_pv_f8_split(const void *q, const int8_t *k, const void *v_image, void *o_part, float *lse_part,
                                           const float *k_scale, const float *v_scale, const float *v_mean,
                                           int B, int Hq, int Hkv, int kv_split, int Lq, int Lk_chunk, int D,
                                           int64_t q_sb, int64_t q_sh, int64_t q_sl, int64_t k_sb, int64_t k_sh, int64_t k_sl,
                                           int64_t o_sb, int64_t o_sh, int64_t o_sl,
                                           int is_causal, float sm_scale_log2, int q_dtype, int out_dtype, void *stream, const SageLaunchAttr *attr);
/* The FP16-PV counterpart (sage_attn_fused_q_pv_f16 over a split key range; v_image is the fp16 image, no v_scale). */
SAGE_API int sage_attn_fused_q_pv_f16_split(const void *q, const int8_t *k, const void *v_image, void *o_part, float *lse_part,
                                            const float *k_scale, const float *v_mean,
                                            int B, int Hq, int Hkv, int kv_split, int Lq, int Lk_chunk, int D,
                                            int64_t q_sb, int64_t q_sh, int64_t q_sl, int64_t k_sb, int64_t k_sh, int64_t k_sl,
                                            int64_t o_sb, int64_t o_sh, int64_t o_sl,
                                            int is_causal, float sm_scale_log2, int q_dtype, int out_dtype, void *stream, const SageLaunchAttr *attr);

/* Merge a partial attention state into a running FP32 state by log-sum-exp (natural log), in place:
 *   m = max(lse_acc, lse_new); w_a = e^(lse_acc-m); w_b = e^(lse_new-m);
 *   o_acc = (o_acc w_a + o_new w_b) / (w_a + w_b);  lse_acc = m + log(w_a + w_b)
 * first != 0 initialises the state from (o_new, lse_new).  o_out (nullable) additionally receives the merged
 * output in the dtype of o_new (pass it on the last step).  o_acc [B,H,L,D] fp32 and lse_* [B,H,L] fp32 are
 * contiguous; o_new / o_out take element strides (HND or NHD views).
 * Replaces: the LSE combine a sequence-parallel caller performs on `sageattn(..., return_lse=True)` results
 * (core.py:782-786,823-826; example/parallel_sageattn_cogvideo.py drives it through xfuser's ring attention). */
SAGE_API int sage_merge_states(float *o_acc, float *lse_acc, const void *o_new, const float *lse_new, void *o_out,
                               int B, int H, int L, int D, int64_t n_sb, int64_t n_sh, int64_t n_sl,
                               int64_t o_sb, int64_t o_sh, int64_t o_sl, int dtype, int first, void *stream);

/* Split-KV merge.  A call whose grid would not fill the chip (few query blocks, long key range: cross-attention, decode-like
 * shapes) is run as S chunks of the key range folded into the batch dimension (plus an optional ragged tail chunk from a
 * second launch); every chunk leaves a normalised partial output (fp16) and its log-sum-exp (log2 domain, the value the
 * attention entry points write to `lse`).  This op combines them in one pass:
 *     m = max_s lse_s;  w_s = 2^(lse_s - m);  o = sum_s w_s o_s / sum_s w_s;  lse = m + log2(sum_s w_s)
 * o_part [B, Hkv, S, group, L, D] fp16 contiguous (H = Hkv * group: the chunks are folded into the kv-head dimension),
 * lse_part [B, Hkv, S, group, L]; o_tail [B, H, L, D] fp16 / lse_tail [B, H, L] nullable; o_out takes element strides;
 * lse_out nullable.
 * Replaces: nothing in the reference's kernels (they parallelise over query blocks only, qk_int_sv_f8_cuda_sm89.cuh:720-738);
 * the combine is the one its sequence-parallel callers apply to `return_lse` results (core.py:782-786). */
SAGE_API int sage_merge_split(const void *o_part, const float *lse_part, const void *o_tail, const float *lse_tail,
                              void *o_out, float *lse_out, int B, int S, int H, int group, int L, int D,
                              int64_t o_sb, int64_t o_sh, int64_t o_sl, int out_dtype, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SAGE_GFX950_H */
