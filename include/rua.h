/*
 * rua.h — C ABI of librua_hip.so: MI355X (gfx950) ragged-sequence row kernels.
 *
 * This is the drop-in boundary for the hot path of speedcell4/torchrua 0.5.1
 * (layout conversion cat/pack/left/right, select head/last/roll/rev/trunc,
 * segmented + scatter reduce).  The reference has no FFI of its own: its
 * boundary is the set of Python methods it monkey-patches onto the four layout
 * types.  Every entry point below cites the reference method(s) whose ATen
 * composition it replaces (paths relative to the reference checkout).
 *
 * Conventions (all entry points):
 *   - plain device pointers + sizes, no torch types; all index vectors int64
 *     (the reference's index dtype everywhere);
 *   - never allocates, never synchronises, never throws; work is enqueued on
 *     `stream` (a hipStream_t passed as void*); scratch is caller-provided;
 *   - returns 0 on success, a positive hipError_t if the HIP runtime refused
 *     the launch, or a negative RUA_E* code for a rejected argument;
 *   - stateless and re-entrant.
 */
#ifndef RUA_H_
#define RUA_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RUA_ABI_VERSION 6

/* argument errors (negative so they cannot collide with hipError_t) */
#define RUA_EINVAL   (-1)  /* bad enum / null pointer / negative size      */
#define RUA_EALIGN   (-2)  /* pointer or row size not aligned as required   */
#define RUA_ERANGE   (-3)  /* size exceeds what the launch geometry covers  */

/* ---- layouts -------------------------------------------------------------
 * Logical coordinate of a token: (b, t), 0 <= t < len[b].  Flat storage row:
 *   CAT    row = off[b] + t                       core/get.py:25-26, layout/cat.py:79-81
 *   LEFT   row = b*T_phys + t                     core/get.py:41-42, layout/left.py:73-77
 *   PACK   row = boff[t] + unsorted[b]            core/get.py:57-58, layout/pack.py:43-45
 *   RIGHT  row = b*T_phys + (T_log - len[b]) + t  core/get.py:73-74, layout/right.py:74-79
 *   LIST   (destination only) row j holds token (bptr[j], tptr[j])   core/get.py tuple keys
 * Enumeration order of a layout's rows (= the order of X.ptr()):
 *   CAT/LEFT/RIGHT: for b: for t < len[b]         layout/cat.py:68-71
 *   PACK:           for t: for r < bsz[t]: (sorted[r], t)            layout/pack.py:23-27
 */
enum rua_kind { RUA_CAT = 0, RUA_LEFT = 1, RUA_PACK = 2, RUA_RIGHT = 3, RUA_LIST = 4 };

typedef struct rua_layout {
  int32_t kind;            /* enum rua_kind */
  int32_t tile_t_log2;     /* PACK with a tile table (below): bits 0-7 log2 of the time steps per tile (4 .. 6),
                              bits 8-15 log2 of the ranks per tile (4); a zero field = 4 (ABI <= 3: 16 x 16);
                              bits 16-23: R = 128 / row bytes (2, 4 or 8) when the table was built for windows that
                              begin up to R - 1 steps early — tile_start[c] counts the ranks alive at step
                              c * TT - (R - 1), and there are ceil((T + R - 1) / TT) chunks — so that the kernel may
                              align every rank's batch-major runs to 128-byte lines; 0 = plain windows;
                              bit 24 (ABI 5): no table — the tiles cover the whole (sequence x step) grid of a padded
                              DESTINATION, ceil(B / ranks) tiles per chunk of steps, n_tchunks = ceil(T_phys / steps):
                              P.left() / P.right() at narrow rows write tokens and fill in one pass;
                              bit 25 (ABI 5): the table counts tiles of ONE time step x (1 << bits 8-15) ranks
                              (n_tchunks = T): a roll inside the PackedSequence moves every step's rows as one run */
  int64_t n_rows;          /* storage rows: CAT/PACK: sum(len); LEFT/RIGHT: B*T_phys; LIST: M */
  int64_t B;               /* number of sequences */
  int64_t T_phys;          /* LEFT/RIGHT: rows per sequence in storage (data.size(1)) */
  int64_t T_log;           /* RIGHT: the T used for right alignment (reference: token_sizes.max()).
                              CAT: optional hint — the longest sequence when the caller knows it, 0 = unknown
                              (rua_enum_rows writes sequence by sequence only when no sequence can be a large
                              share of the launch; results never depend on it) */
  const int64_t* lens;     /* [B] or NULL                                   */
  int64_t len_add;         /* len[b] = (lens ? lens[b] : 0) + len_add       */
  const int64_t* off;      /* CAT: exclusive scan of lens, [B] (NULL when lens is NULL);
                              effective offset = off[b] + b*len_add         */
  const int64_t* boff;     /* PACK: exclusive scan of batch_sizes, [T]      */
  int64_t T;               /* PACK: number of time steps                    */
  const int64_t* sorted;   /* PACK: sorted_indices [B]                      */
  const int64_t* unsorted; /* PACK: unsorted_indices [B]                    */
  const int64_t* bptr;     /* LIST: [M]; NULL = all zeros (tptr then indexes ONE sequence,
                              e.g. a flat row gather `data[key]` against LEFT{B=1}; a negative
                              entry of such a flat list wraps by the source's n_rows, like
                              torch's own indexing in core/get.py:29, core/set.py:30)   */
  const int64_t* tptr;     /* LIST: [M]                                     */
  /* PACK, optional: a (rank x time) tile table that lets narrow-row C/L/R <-> P transposes move
   * multi-row runs on BOTH sides (rua_move_rows picks it up when rows are <= 64 bytes) .        */
  const int64_t* bsz;        /* PACK: device copy of batch_sizes [T].  CAT (optional, ABI 6): ONE word, the number of
                                lengths <= 0 (rua_exclusive_scan_i64's total[1]): 0 lets max / min / logsumexp skip
                                the tracking of the reference's global `initial` (rua_segment_reduce)            */
  const int64_t* tile_start; /* [n_tchunks + 1]: tile_start[c] = sum_{c'<c} ceil(batch_sizes[TT*c']/TR) (TT time steps, TR ranks per tile) */
  int64_t n_tchunks;         /* ceil(T / TT)                                                     */
  int64_t n_tiles;           /* tile_start[n_tchunks] (the caller knows it: batch_sizes is a host tensor) */
} rua_layout;

/* ---- per-sequence token maps: t_src = f(t_dst) ---------------------------- */
enum rua_tmap {
  RUA_T_SHIFT = 0,  /* t + arg            identity (arg=0), trunc (select/trunc.py:9-62)       */
  RUA_T_ROLL  = 1,  /* (t - arg) mod slen select/roll.py:6-13                                  */
  RUA_T_REV_S = 2,  /* slen - 1 - t       select/rev.py:6-41; last = REV_S with dlen = 1 (select/last.py:7-13) */
  RUA_T_REV_D = 3,  /* dlen - 1 - t       adjoint of REV_S (backward of last)                  */
  RUA_T_ZERO  = 4   /* 0                  broadcast one row per sequence (backward of sum)     */
};

#define RUA_MOVE_SCATTER 1  /* flags: enumerate `dst` layout rows as the SOURCE rows and write
                               them to the rows computed from `src` layout (core/set.py)        */
#define RUA_MOVE_NT_ON   2  /* force / forbid non-temporal payload accesses; default: on when the   */
#define RUA_MOVE_NT_OFF  4  /* destination is >= 512 MiB (cannot stay in the 256 MiB Infinity Cache) */
/* launch geometry of the mover (default: ~16 KiB of destination rows per workgroup, one contiguous span of tiles
 * per XCD on big launches; these override it, for A/B runs):
 * bits 4-7 = log2 of the destination rows one workgroup takes (2..8 -> 4..256 rows), */
#define RUA_MOVE_TILE_LOG2(k) (((k) & 0xf) << 4)
#define RUA_MOVE_XCD_SPAN_ON  256  /* every XCD takes ONE contiguous span of tiles (blockIdx % 8 picks the span) */
#define RUA_MOVE_XCD_SPAN_OFF 512  /* tiles in plain blockIdx order                                             */
#define RUA_MOVE_NO_TAIL8 1024     /* rows of 8 (mod 16) bytes: keep 8-byte lanes instead of 16-byte lanes + an 8-byte tail */
#define RUA_MOVE_NO_NARROW 2048    /* rows of one vector (1 .. 16 bytes): the generic kernel instead of the lane-per-row one */

/* K1. Exclusive prefix sum of n int64 (wavefront scan).  out[i] = sum(in[0..i)).
 * `ws` must hold rua_scan_ws_elems(n) int64.  If total != NULL it points at TWO device words (ABI 6; one before):
 * total[0] = sum(in), total[1] = #{i : in[i] <= 0} — of a length vector: the number of EMPTY sequences, which a CAT
 * layout may hand to rua_segment_reduce (rua_layout::bsz).
 * Replaces get_offsets, utils.py:16-19 (cumsum + roll + [0]=0). */
int64_t rua_scan_ws_elems(int64_t n);
int rua_exclusive_scan_i64(const int64_t* in, int64_t* out, int64_t* total, int64_t n,
                           int64_t* ws, void* stream);

/* K3. PackedSequence metadata from lengths + the (host-sorted) descending order.
 *   unsorted[sorted[r]] = r                       utils.py:22-26 (invert_permutation)
 *   bsz[t] = #{b : len[b] > t}, t < T             core/view.py:47-58 (get_mask(..).sum(dim=0))
 * without materialising the B x T int64 mask (core/view.py:11-18). */
int rua_pack_meta(const int64_t* lens, const int64_t* sorted, int64_t B, int64_t T,
                  int64_t* unsorted, int64_t* bsz, void* stream);

/* K3 + K1 in one call — everything pack() derives on the device: rua_pack_meta's outputs plus
 *   boff[t] = sum(bsz[0..t))  (T entries)   and, if off != NULL,   off[b] = sum(lens[0..b))  (B entries).
 * Moderate sizes (T <= 2 048, B <= 131 072) take ONE launch instead of five; larger ones run the three steps back to
 * back.  `ws`: rua_scan_ws_elems(max(B, T)) int64 (only touched on the large path).  core/view.py:47-58 + utils.py:16-19. */
int rua_pack_prepare(const int64_t* lens, const int64_t* sorted, int64_t B, int64_t T, int64_t* unsorted,
                     int64_t* bsz, int64_t* boff, int64_t* off, int64_t* ws, void* stream);

/* K3b. token_sizes of a PackedSequence in original batch order:
 *   len[b] = #{t : bsz[t] > unsorted[b]}          core/view.py:21-25 (get_mask(P).sum(dim=1)) */
int rua_lens_from_pack(const int64_t* bsz, int64_t T, const int64_t* unsorted, int64_t B,
                       int64_t* lens, void* stream);

/* K2. Enumerate a layout's rows in ptr() order.  Any of the outputs may be NULL.
 *   batch_ptr[j], token_ptr[j]                    utils.py:7-13 (major_sizes_to_ptr), X.ptr()
 *   flat[j] = storage row of token j              layout/left.py:73-77, right.py:74-79 (X.idx()) */
int rua_enum_rows(const rua_layout* lay, int64_t n_tokens, int64_t* batch_ptr, int64_t* token_ptr,
                  int64_t* flat, void* stream);

/* get_mask / mask: out[b, t] = (t < len[b]) ? one : zero over a B x T grid of elem_bytes-wide
 * elements (1, 2, 4 or 8).  core/view.py:11-18, mask.py:6-14. */
int rua_mask(const int64_t* lens, int64_t B, int64_t T, void* out, int32_t elem_bytes,
             uint64_t zero_bits, uint64_t one_bits, void* stream);

/* K4/K5/K6/K7. The row mover.  For every storage row j of `dst` (token (b,t), or padding):
 *     ts = tmap(t);  if 0 <= ts < slen[b]: dst_row(j) = src_row(b, ts)  else  dst_row(j) = fill
 * Replaces every conversion in core/cast.py:8-71 (+ the new_full pre-fill of core/view.py:34-38,
 * 67-71: padding and payload are written in ONE pass), core/get.py / core/set.py tuple-key
 * indexing, select/head.py, select/last.py, select/roll.py, select/rev.py, select/trunc.py.
 * `fill16` is the fill element replicated to 16 bytes. Rows are row_bytes wide on both sides.
 * `pad_row` >= 0 makes PADDING rows of a LEFT/RIGHT destination copies of that source storage
 * row instead of `fill16` (select/roll.py:19-23,33-37: the reference pads its index tensor with
 * 0, so its padding rows come out as copies of storage row 0); -1 = use the fill. */
int rua_move_rows(const rua_layout* dst, const rua_layout* src, int32_t tmap, int64_t tmap_arg,
                  void* dst_data, const void* src_data, int64_t row_bytes,
                  const void* fill16, int64_t pad_row, int32_t flags, void* stream);

/* ---- reductions ------------------------------------------------------------ */
enum rua_dtype {
  RUA_F32 = 0, RUA_BF16 = 1, RUA_F16 = 2, RUA_F64 = 3,
  /* integer element types (ABI 4): rua_segment_reduce only, over a CAT layout with or without `perm` — the
   * scatter_* of reduce.py:6-23 on integer tensors, which the reference hands to torch.index_reduce / index_add like
   * any other dtype.  SUM / MEAN / MAX / MIN / PROD in the element type itself, bit-exact: sums and products wrap,
   * MEAN is ATen's floor division by a count held in the same type (so the count wraps too); `extreme`, `ws`,
   * `ties_out` and `empty_bits` are ignored (empty sequences yield the op's identity unless include_self == 2),
   * LOGSUMEXP is RUA_EINVAL. */
  RUA_I64 = 4, RUA_I32 = 5, RUA_I16 = 6, RUA_I8 = 7, RUA_U8 = 8
};
#define RUA_TIES_FINAL 2   /* rua_segment_reduce_backward's include_self: see there */
#define RUA_BWD_FILL_PADDING 0x100  /* OR-ed into that include_self: also write zeros into the rows of a padded
                                      layout that hold no token (grad_in then needs no pre-zeroing)              */
#define RUA_BWD_TIES_POSITIVE 0x200 /* OR-ed into that include_self (MAX / MIN): tied extrema share a POSITIVE
                                      gradient (g / ties each) and each take a non-positive one WHOLE — what
                                      torch.segment_reduce's backward does (segment_max/min, reduce.py:34-41);
                                      without it ties share g / ties whatever the sign (index_reduce's backward:
                                      scatter_max/min, reduce.py:6-11)                                          */
enum rua_op {
  RUA_SUM = 0, RUA_MEAN = 1, RUA_MAX = 2, RUA_MIN = 3, RUA_PROD = 4, RUA_LOGSUMEXP = 5
};

/* K8/K9/K10/K11. out[b, :] = op over t < len[b] of data[row(b, t), :], accumulating in fp32
 * (fp64 for RUA_F64), for the sequences of `lay` in ANY layout:
 *   CAT   = torch.segment_reduce(data, op, lengths=lens)      reduce.py:34-61
 *   PACK  = the same over a PackedSequence without P.cat()     (core/cast.py:8-10 + reduce.py:44)
 *   LEFT/RIGHT = over the valid rows of a padded batch          segment.py:16-25, 38-47
 * `perm` (may be NULL) indirects CAT rows: row = perm[off[b]+t] — the sorted-by-destination
 * form of scatter_* (reduce.py:6-31).
 * Empty sequence -> `empty_bits` (the reference's `initial`: 0, 1, or the global min/max).
 * If `extreme` != NULL (MAX/MIN/LOGSUMEXP; RUA_EXTREME_WORDS uint64 of scratch, initialised by the library) the call
 * reproduces the reference's `initial = tensor.min()` / `.max()` (reduce.py:35,40,57) without its extra pass over the
 * data: every wave of the reduce folds the OPPOSITE extreme of the rows it reads into one of the 1 024 hashed slots
 * extreme[0..1023] (one atomic per wave) and raises flags in extreme[1024] — bit 0: some element is NaN (then `initial`
 * is NaN and poisons every segment), bit 1: some segment is empty.  rua_fill_empty then patches the output, every workgroup its share; with no NaN and no
 * empty segment (the common case) its workgroups read one word and leave.  (ABI <= 5 took a second walk over the
 * payload when a segment was empty; ABI 6 never reads the payload twice.)
 * include_self: 0 overwrite | 1 `out` already holds values that take part (scatter_* include_self) |
 *               2 rows of empty sequences are left untouched (torch.index_reduce semantics).
 * split_rows > 0 (with `ws` of rua_reduce_ws_bytes(lay->n_rows, H, dtype, split_rows) bytes) cuts sequences
 * longer than split_rows into parts handled by separate waves (published through a device-side work list,
 * fp32 partials folded in part order: deterministic); 0 = one wave streams each sequence.
 * Integer dtypes (RUA_I64 .. RUA_U8; CAT only, SUM / MEAN / MAX / MIN / PROD in the element type, as ATen's
 * index_reduce / index_add do): the same two arguments cut buckets longer than split_rows by POSITION into ranges of
 * split_rows, int64 partials in `ws` (exact in any order; two small launches when no bucket is long).
 * ties_out (MAX/MIN; may be NULL): [B, H] f32 (f64 for RUA_F64) that receives, per output element, how many elements
 * of the sequence equal it (with include_self == 1 the old row is folded into the result but not counted; rows that
 * include_self == 2 leaves untouched are not written: pre-zero the buffer) — what the backward needs, for free in the pass that
 * reads the payload anyway (rua_segment_reduce_backward with include_self = RUA_TIES_FINAL then takes ONE walk).
 * Bits that may be OR-ed into `op`:
 *   RUA_OP_SCRATCH_CLEAN  (here, in rua_pack_reduce and in rua_fill_empty) by a caller that keeps ONE persistent `extreme`
 *                         scratch of RUA_EXTREME_WORDS uint64 per stream, zeroed once when it was allocated: the scratch
 *                         arrives zeroed, so no initialising launch; rua_fill_empty (which must then be called with the
 *                         same bit) hands it back zeroed.
 *   RUA_OP_NO_EMPTY       (here and in rua_pack_reduce) the caller PROVES that no sequence is empty — lengths it holds on
 *                         the host, a PackedSequence whose batch_sizes[0] equals its sequence count: nothing will ever
 *                         ask for the global extreme, so the reduce does not track it (a NaN still poisons everything:
 *                         that needs no extreme).  Up to ABI 5 the bit meant "do not arm the second walk".  The device
 *                         can prove the same by itself: a CAT layout may carry in `bsz` a pointer to the number of
 *                         lengths <= 0, as rua_exclusive_scan_i64 leaves it in total[1] (rua_layout, below).
 * (Dropping the trailing launch altogether — the reduce's last wave patching the output, found by tickets — was built
 * and measured in round 5: no gain at the BASELINE shapes, a loss where waves are short; profiles/r05_self_patch_ab.txt.) */
#define RUA_EXTREME_WORDS    1027
#define RUA_OP_SCRATCH_CLEAN 0x100
#define RUA_OP_NO_EMPTY      0x200
/* rua_segment_reduce over a CattedSequence with rows narrower than 1 KiB: the caller KNOWS the lengths and vouches
 * that no sequence is far above the average (torchrua_amd: at most 8 x the average, or 64 rows).  When the sequences
 * are short (16 .. 64 rows on average by row width) every row slot of a wave then takes a sequence of its own — one
 * wave = one workgroup per sequence is bound by the workgroup dispatch rate there — and the wave walks to the longest of
 * them, hence the word.  (At rows of <= 32 bytes four sequences share a wave with or without it: that form checks its
 * own lengths, wave by wave.  Since ABI 6 so does every-row-slot-its-own-sequence when the word is NOT given — a batch
 * is short on average whatever the caller knows: rows / sequences — and a wave whose lengths are far apart walks them
 * one after the other; the word saves that check.)  A hint: results do not depend on it (sums to rounding: the fold's
 * association changes). */
#define RUA_OP_SHORT_SEQS    0x400
int64_t rua_reduce_ws_bytes(int64_t n_rows, int64_t H, int32_t dtype, int64_t split_rows);
/* Waves (1, 2 or 4) that share one sequence in rua_segment_reduce for an aligned payload of `row_bytes`-wide
 * rows (multiples of 16 bytes, or of 8 bytes beyond one vector), B sequences, n_rows rows in all — the launcher's own rule, exported so that a host planner pricing
 * `split_rows` (a unit streams team-times as fast) cannot drift from it. */
int rua_reduce_team_waves(int64_t n_rows, int64_t B, int64_t row_bytes);
int rua_segment_reduce(const rua_layout* lay, const int64_t* perm, const void* data, void* out,
                       int64_t H, int32_t dtype, int32_t op, int32_t include_self,
                       uint64_t empty_bits, void* extreme, int64_t split_rows, void* ws, void* ties_out,
                       void* stream);

/* Fused pack + reduce (an EXTENSION: the reference has no one-call equivalent; it is exactly
 * core/cast.py:41-49 followed by the reduction of reduce.py:34-61 over the packed rows).  One pass over
 * the payload of `src` (CAT/LEFT/RIGHT): every row is stored to its row of the PackedSequence `pack`
 * (boff[t] + unsorted[b]) AND folded into out[b, :].  The same PackedSequence, bit for bit, as rua_move_rows(pack <- src);
 * the same fp32 accumulation as rua_segment_reduce(pack) (bit-identical whenever that call gives each sequence to one wave —
 * few-but-long batches go through a team of waves there, which associates the partial sums differently), at 2/3 of the HBM traffic.  Needs H*sizeof(dtype) % 16 == 0 and 16-byte
 * aligned pointers (returns RUA_EALIGN otherwise: run the two-call form). */
int rua_pack_reduce(const rua_layout* src, const rua_layout* pack, const void* data, void* pack_data, void* out,
                    int64_t H, int32_t dtype, int32_t op, uint64_t empty_bits, void* extreme, int64_t split_rows,
                    void* ws, void* stream);

/* Backward of rua_segment_reduce in one kernel (semantics of torch's
 * segment_reduce backward, which the reference inherits through autograd: reduce.py:34-61):
 *   SUM g | MEAN g/len | PROD g*prod(others) (zero factors handled like torch) | LOGSUMEXP g*exp(x-out) |
 *   MAX/MIN g/ties where x == out, else 0 (RUA_BWD_TIES_POSITIVE: g itself when g <= 0 or NaN).
 * grad_in has the storage of `data`; rows of padded layouts that hold no token are NOT written unless
 * RUA_BWD_FILL_PADDING is OR-ed into include_self (SUM / MEAN / LOGSUMEXP and MAX / MIN with RUA_TIES_FINAL then
 * write them in the same pass: the backward runs one storage row at a time, laid out like rua_move_rows).
 * With `perm` it is the gradient w.r.t. the SOURCE rows of scatter_* (reduce.py:6-31); include_self != 0 then
 * counts the old destination row in MEAN's divisor (MAX/MIN: seed `ties` with the old row's tie, below).
 * split_rows / ws as in rua_segment_reduce.
 * ties (MAX/MIN; may be NULL): [B, H] accumulators (f32, f64 for RUA_F64) that the caller pre-sets to the ties
 * the rows of `data` do not see (0, or 1 where the old destination row of a scatter_max/min with include_self
 * equals `out`).  The kernel adds every sequence's own ties (integer-valued float atomics: exact), then divides
 * the gradient by the total — which lets long sequences be split, and leaves the totals for the caller.
 * With ties == NULL each sequence is counted and applied by one wave (no splitting for MAX/MIN).
 * include_self == RUA_TIES_FINAL: `ties` already holds the complete counts (the forward's ties_out): no counting walk.
 * self_in (perm != NULL only; may be NULL): the old destination rows [B, H] of a scatter_* — the `tensor` argument of
 * reduce.py:6-31.  MAX/MIN with RUA_TIES_FINAL: the kernel itself adds the tie of the old row where it equals `out`
 * (torch's index_reduce backward counts it with and without include_self), so `ties` is the forward's ties_out
 * unchanged.  PROD with include_self == 1: the old row is one more factor of every source row's gradient
 * (g * tensor * prod(other sources), zero factors handled like torch). */
int rua_segment_reduce_backward(const rua_layout* lay, const int64_t* perm, const void* data, const void* out,
                                const void* grad_out, void* grad_in, int64_t H, int32_t dtype, int32_t op,
                                int32_t include_self, int64_t split_rows, void* ws, void* ties,
                                const void* self_in, void* stream);

/* Gradient of scatter_* w.r.t. the destination `tensor` (reduce.py:6-31; torch's index_add / index_reduce backward),
 * one elementwise launch over [S, H]:
 *   include_self != 0:  SUM g | MEAN g/(counts+1) | MAX/MIN (tensor == out) ? g/(aux+1) : 0 | PROD g*aux |
 *                       LOGSUMEXP g*exp(tensor - out)
 *   include_self == 0:  g in rows no index names (counts == 0: they keep `tensor`), 0 elsewhere.
 * counts[S]: bucket sizes (rua_index_buckets).  aux [S, H]: MAX/MIN — the source rows' tie counts (f32; f64 for
 * RUA_F64; rua_segment_reduce's ties_out; NULL = none); PROD — the product of each bucket's source rows in the
 * payload dtype (rua_segment_reduce into ones with include_self = 2).  self_in/out may be NULL where unused. */
int rua_scatter_self_grad(const int64_t* counts, int64_t S, int64_t H, const void* self_in, const void* out,
                          const void* grad_out, const void* aux, void* grad_self, int32_t dtype, int32_t op,
                          int32_t include_self, void* stream);

/* After rua_segment_reduce / rua_pack_reduce with `extreme` (MAX/MIN/LOGSUMEXP): write the
 * global extreme — the reduce left it in the scratch — into the rows of empty sequences, or NaN into every row when
 * the NaN flag is up (the reference's initial=NaN behaviour).  Every workgroup patches its share of the batch.
 * (ABI <= 5 also took `data` and `perm` for a second walk over the payload; there is none any more.) */
int rua_fill_empty(const rua_layout* lay, void* out, int64_t H, int32_t dtype, int32_t op,
                   void* extreme, void* stream);

/* Bucket `index` (values in [0,S); others are ignored): counts[S], off[S] (exclusive scan) and perm[M] such that
 * perm[off[s] .. off[s]+counts[s]) are the rows i with index[i] == s IN ASCENDING ORDER — a stable radix sort on the
 * destination, deterministic for any fan-in.  513 .. 262 144 destinations with M < 2^31 (ABI 4): most significant
 * digit first in two levels, the second local to a bin, 32-bit words where (low digit, row) fit them
 * (rua_bucket_msd.hip; entries whose index is out of range are dropped and the tail of `perm` behind the valid
 * entries is left unwritten); otherwise least significant digit first over packed (destination, row) words with
 * <= 9-bit digits (rua_scatter.hip; needs bits(S) + bits(M) <= 62, RUA_ERANGE otherwise).  `ws` holds
 * rua_bucket_ws_elems(M, S) int64.  Feeds rua_segment_reduce(perm=..) for scatter_* (reduce.py:6-31). */
int64_t rua_bucket_ws_elems(int64_t M, int64_t S);
int rua_index_buckets(const int64_t* index, int64_t M, int64_t S, int64_t* counts, int64_t* off,
                      int64_t* perm, int64_t* ws, void* stream);

/* ---- host side of pack() (no GPU involved; plain host pointers) ------------------------------------------
 * sorted_indices of core/view.py:48 — `torch.sort(token_sizes.cpu(), descending=True)` — for int64 keys, with the
 * SAME tie order: ATen's CPU kernel is the C++ library's introsort over (key, index) pairs compared by key only,
 * a deterministic function of the input that is reproduced here step for step, the two halves of every partition
 * on different threads (n_threads >= 1, the caller included).  The Python layer verifies the equality against
 * torch.sort itself at first use and otherwise keeps making the reference's call. */
int rua_host_sort_desc(const int64_t* keys, int64_t n, int64_t* sorted_indices, int32_t n_threads);
/* The same sort on a helper thread (ABI 4): `begin` returns at once, `end` waits for the job and returns its code;
 * keys / sorted_indices must stay valid in between, one job at a time (RUA_EINVAL if one is already posted, or if `end`
 * finds none).  pack() with device-only lengths uses the interval for the rest of its host work (core/view.py:47-58:
 * batch_sizes, the offset scans), because the GPU idles until the order is known. */
int rua_host_sort_desc_begin(const int64_t* keys, int64_t n, int64_t* sorted_indices, int32_t n_threads);
int rua_host_sort_desc_end(void);
/* diagnostics for the tests: how many segments, over all calls so far, exhausted the introsort's depth budget and
 * were heap-sorted (the branch a random input never reaches; tests/golden/sort_killer.npy does) */
int64_t rua_host_sort_heap_segments(void);

/* batch_sizes[t] = #{b : lens[b] > t}, t < T — the CPU tensor PackedSequence mandates
 * (core/view.py:55: get_mask(self).sum(dim=0).cpu()), from the host copy of the lengths. */
int rua_host_batch_sizes(const int64_t* lens, int64_t B, int64_t T, int64_t* batch_sizes);

/* The two exclusive scans pack() needs next to batch_sizes (layout/pack.py:43-45 `offsets()`, utils.py:16-19), on the
 * host: boff[t] = sum of batch_sizes[< t], off[b] = sum of lens[< b] (either may be NULL).  With device-only lengths
 * the host computes them while the order is being sorted and uploads them with it (one launch less in front of the
 * mover); with a host mirror of the lengths they are derived on the device (rua_pack_prepare). */
int rua_host_pack_scans(const int64_t* lens, int64_t B, const int64_t* batch_sizes, int64_t T, int64_t* boff,
                        int64_t* off);

/* Introspection: ABI version and the gfx target the code objects were built for. */
int rua_abi_version(void);
const char* rua_build_target(void);

#ifdef __cplusplus
}
#endif
#endif /* RUA_H_ */
