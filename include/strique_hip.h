/*
 * libstrique_hip -- C ABI of the MI355X (gfx950) implementation of STRique's per-read hot path.
 *
 * Plain C, caller-owned buffers, no torch / pybind types.  Every function returns 0 on success
 * or a STRQ_ERR_* code; strq_last_error(ctx) gives the message.  One ctx per process per GPU;
 * calls on one ctx are not re-entrant (the reference holds the GIL for the whole native call,
 * src/pyalign.cpp:59-61, and each worker process owns its own aligner, scripts/STRique.py:743-745).
 *
 * What each entry point replaces in giesselmann/STRique (file:line in the reference tree):
 *   strq_ctx_create / strq_ctx_destroy   pyseqan.align_raw()            src/pyalign.cpp:50
 *   strq_set_align_params / get          the 8 float properties         src/pyalign.cpp:51-58,
 *                                                                       src/align_raw.h:84-103
 *   strq_align_overlap                   align_raw.align_overlap(a, b)  src/pyalign.cpp:59-61,
 *                                                                       src/align_raw.h:106-158
 *   strq_model_create / strq_viterbi     pomegranate HiddenMarkovModel.bake()/.viterbi() as used
 *                                        at scripts/STRique.py:431,434,490,493
 *   strq_target_add / strq_detect_batch  repeatCounter.add_target / detect
 *                                                                       scripts/STRique.py:553-618
 */
#ifndef STRIQUE_HIP_H
#define STRIQUE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STRQ_OK 0
#define STRQ_ERR_ARG 1          /* bad argument */
#define STRQ_ERR_DEVICE 2       /* HIP runtime / no GPU */
#define STRQ_ERR_UNSUPPORTED 3  /* input outside what the kernels cover (documented per call) */
#define STRQ_ERR_NOMEM 4

typedef struct strq_ctx strq_ctx;

/* Version of this ABI (bumped on any signature change). */
int strq_abi_version(void);   /* currently 12 (12: strq_batch_fetch_range, strq_last_overlap, two sub-batches in flight; 11: strq_set_option, strq_get_option, strq_batch_upload_part, strq_last_screen_mode; 10: strq_last_screen, strq_debug_screen_plan; 9: strq_last_viterbi_launches, strq_last_second_round, strq_inflate_backend, strq_inflate_many, strq_inflate_stats, strq_h5_locate, strq_vbz_chunks; 8: strq_model_set_positions; 5: strq_host_stats, host_stats of strq_detect_batch / strq_batch_upload optional; 6: strq_detect_batch_reads; 7: strq_last_geometry, strq_batch_run_range, strq_inflate_chunks) */

/* Create a context on HIP device `device_id`.  Fails (STRQ_ERR_DEVICE) when no GPU is present:
 * there is no CPU fallback in this library. */
int strq_ctx_create(int device_id, strq_ctx** out);
void strq_ctx_destroy(strq_ctx* ctx);
const char* strq_last_error(const strq_ctx* ctx);
/* hipDeviceSynchronize on the context's device (benchmark brackets). */
int strq_device_synchronize(strq_ctx* ctx);

/* Switches of the library (the experiment / A-B switches that used to be environment variables only, e.g. STRQ_NO_SCREEN,
 * STRQ_OVERLAP, STRQ_SEG; README "Switches"): per context when ctx != NULL, process-wide when ctx == NULL.  Every switch is read
 * in this order: context, process, environment variable of the same name.  value NULL removes the entry (back to the next level),
 * "" means "not set" whatever the lower levels say.  Keys start with "STRQ_".  No switch changes a result -- they choose between
 * kernels / plans that are all exact (the reference has no counterpart: scripts/STRique.py:904-914 exposes --t and --config only).
 * strq_get_option writes the effective value ("" when unset) to out[cap]. */
int strq_set_option(strq_ctx* ctx, const char* key, const char* value);
int strq_get_option(const strq_ctx* ctx, const char* key, char* out, int32_t cap);

/* Alignment parameters, order: open_h, ext_h, open_v, ext_v, dist_offset, dist_min.
 * Defaults after create are align_raw's own (src/align_raw.h:51-60): -2,-8,-2,-8, 8,-16;
 * STRique overrides them from its config (scripts/STRique.py:507-523). */
int strq_set_align_params(strq_ctx* ctx, const float params[6]);
int strq_get_align_params(const strq_ctx* ctx, float params[6]);

/*
 * Semi-global alignment of flank `b` (m samples, end to end) inside read `a` (n samples, free
 * ends) -- align_raw.align_overlap(a, b).
 *   score      best score (float32, as the reference returns it)
 *   a_idx[n]   view position of every element of a   (nullable)
 *   b_idx[m]   view position of every element of b   (nullable)
 *   rec[m]     compact per-flank-row record (nullable): (j << 1) | is_vertical, where for a
 *              diagonal step b[k] is aligned to a[j-1] and for a vertical step j samples of `a`
 *              precede b[k]
 *   j_end,j0   DP columns where the path ends / leaves the free top row (nullable)
 * Any float32 inputs are accepted, like the reference (src/pyalign.cpp:59-61).  What
 * repeatCounter.detect passes -- `a` with at most 256 distinct values (an 8-bit morphology signal) and `b`
 * made of runs of 6 equal samples, m <= 49152 (scripts/STRique.py:592-601,562-565) -- runs on the
 * LDS-table wavefront kernels; everything else on a generic kernel (one trace byte per cell in HBM like
 * the reference: (n + 1) x (m + 1) <= 1.7e10, at most 2e9 distinct (a, b) value pairs), same results.
 */
int strq_align_overlap(strq_ctx* ctx, const float* a, int64_t n, const float* b, int64_t m,
                       float* score, uint64_t* a_idx, uint64_t* b_idx,
                       int32_t* rec, int64_t* j_end, int64_t* j0);

/*
 * Batched form of the same alignment for 8-bit level signals (the throughput path).
 *   n_align            number of alignments
 *   levels             concatenated level streams (uint8), one per *read*
 *   read_off[n_reads+1]offsets of each read in `levels`
 *   level_val          n_reads x 256 float32: value of each level of each read
 *   align_read[n_align]read index of each alignment
 *   flank              concatenated flank templates (float32, runs of `samples` equal values)
 *   flank_off[n_align+1]
 *   samples            run length of the flank templates (6)
 * Outputs per alignment: score, j_end, j0 and rec (concatenated like `flank`).
 */
int strq_align_batch(strq_ctx* ctx, int64_t n_align, int64_t n_reads,
                     const uint8_t* levels, const int64_t* read_off, const float* level_val,
                     const int32_t* align_read, const float* flank, const int64_t* flank_off,
                     int32_t samples,
                     float* score, int64_t* j_end, int64_t* j0, int32_t* rec);

/*
 * Upload a "baked" profile HMM (what pomegranate's HiddenMarkovModel.bake(merge='All') leaves,
 * scripts/STRique.py:431,490): emitting states first, silent states after them in topological
 * order, in-edges in CSR form sorted by source.
 *   emis_kind  1 = Normal(mu, sigma):  a = mu, b = 1/(2 sigma^2), c = -log(sigma sqrt(2 pi))
 *              2 = Uniform(lo, hi):    a = lo, b = hi,            c = -log(hi - lo)
 *   count_inc  n_states ints (nullable): states whose visits along the best path are counted
 *              (the two dummy states of repeatHMM, scripts/STRique.py:374-378)
 *   state_tag  n_states ints (nullable): 1 = state belongs to the repeat section (flanked model,
 *              `'repeat' in name`, STRique.py:608) or to the modified branch (modification model,
 *              `'mod' in name`, STRique.py:497); 2 = hub state s0/e0 of the modification model
 *   hint_slot / hint_lane  n_states ints each (nullable): where to keep an emitting state on the
 *              wave -- states of one profile column type in one slot, lane = profile position -- so
 *              that the LDS reads of neighbouring lanes are contiguous.  Purely a performance hint:
 *              an invalid or missing hint falls back to an automatic placement, results are identical.
 * Models of up to 512 emitting and 256 silent states with at most 8 in-edges per state run on the wave-per-window
 * kernels; anything beyond, up to 4096 states, on a general (slower) kernel; larger ones return STRQ_ERR_UNSUPPORTED.
 */
int strq_model_create(strq_ctx* ctx, int32_t n_states, int32_t silent_start, int32_t start, int32_t end,
                      const int32_t* in_ptr, const int32_t* in_src, const double* in_logp,
                      const int32_t* emis_kind, const double* emis_a, const double* emis_b,
                      const double* emis_c, const int32_t* count_inc, const int32_t* state_tag,
                      const int32_t* hint_slot, const int32_t* hint_lane, int32_t* model_id);

/*
 * Optional, after strq_model_create: where every emitting state of a profile-HMM chain sits along the chain --
 * kind[e] 0 for a match-type state, 1 for an insert-type state, pos[e] >= 0 its position (prefix profile, repeat unit,
 * the two dummy states at one position, suffix profile: scripts/STRique.py:236-300,313-354,401-417 laid end to end).
 * A model that is such a chain (every in-edge joins a position with itself or the one before, plus the two edges that close
 * the repeat loop) additionally gets a register-resident image: count / mark decodes then keep the value vector in VGPRs
 * and touch no LDS.  Purely a performance hint: results are identical; STRQ_ERR_UNSUPPORTED (strq_last_error says why)
 * when the model is not such a chain -- it stays valid and runs on its lane layout.
 */
int strq_model_set_positions(strq_ctx* ctx, int32_t model_id, const int32_t* kind, const int32_t* pos);

/*
 * HiddenMarkovModel.viterbi(x) (scripts/STRique.py:434,493).
 *   logp     log-probability of the best path (-inf and status 1 when there is none)
 *   counted  visits of count_inc states on that path
 *   path[T]  emitting state (index in the baked order) of every observation (nullable; asking
 *            for it makes the kernel write back-pointers)
 */
int strq_viterbi(strq_ctx* ctx, int32_t model_id, const double* x, int64_t T,
                 double* logp, int64_t* counted, int32_t* status, int32_t* path);
int strq_viterbi_batch(strq_ctx* ctx, int32_t model_id, int64_t n_seq, const double* x,
                       const int64_t* x_off, double* logp, int64_t* counted, int32_t* status,
                       int32_t* paths);

/*
 * ---- repeatCounter.add_target / detect (scripts/STRique.py:553-618) as one device pipeline ----
 *
 * strq_set_pore_stats   the four model-side constants of pore_model.normalize2model('minmax')
 *                       (STRique.py:154,157-158,123-126): medians of the k-mer means below the 1st /
 *                       above the 99th percentile, model_min, model_max.
 * strq_target_add       one strand of one target (the reference's target_classifier tuple,
 *                       STRique.py:560-576): the two full flank templates generate_signal(*_ext,
 *                       samples) as float32, trim_prefix = len(prefix_ext) - len(prefix) and
 *                       trim_suffix likewise (STRique.py:598-599), the flanked-repeat HMM uploaded
 *                       with strq_model_create and count_bias = flanking_count - repeat_offset
 *                       (STRique.py:374-378,412,437).  samples: the run length of the templates (6 is
 *                       what STRique ships; any value works, on slower kernel shapes).  A template of up
 *                       to 8192 k-mer classes (49152 samples at samples = 6: a flank of 8197 nt; the
 *                       bundled loci have 150 nt = 870 samples) -- longer ones return STRQ_ERR_UNSUPPORTED.
 * strq_detect_batch     detect() for n_reads reads.  signals: concatenated raw samples,
 *                       dtype 0 = int16 (fast5 DAC values), 1 = float64 (pA, as the reference's
 *                       unit tests feed them).  For int16 every order statistic of the conditioning
 *                       comes from exact histograms on the GPU and host_stats is ignored.  float64
 *                       reads have no histogram: their six scalars per read -- median and MAD of the
 *                       median-filtered signal, (c1, h1) of its minmax map and of the raw signal's
 *                       (STRique.py:142-143,152-160,590-592) -- come from a radix selection over the
 *                       samples and numpy's own summation tree on the GPU when host_stats is NULL
 *                       (cond_kernels.hip: f64_stats_kernel; bit-equal to np.median / np.mean /
 *                       np.percentile), or from the caller (six doubles per read, e.g. strq_host_stats).
 * A read whose normalisation is undefined (constant signal, empty percentile tails: numpy hands the
 * reference NaN medians there) gets status 1 and the n = 0 row the reference writes for it -- its
 * offset / ticks come from aligning an all-NaN signal, every cell scoring dist_min, which is what
 * align_overlap makes of it (src/align_raw.h:100); it never aborts the batch.
 */
typedef struct strq_result {
    int32_t count;          /* repeat count n (0 if the gate failed, STRique.py:602-603) */
    int32_t status;         /* 0 ok, 1 signal could not be normalised */
    double score_prefix, score_suffix, log_p;
    int64_t offset, ticks;  /* prefix_end, max(suffix_begin - prefix_end, 0) (STRique.py:616) */
    int64_t prefix_begin, prefix_end, suffix_begin, suffix_end;
} strq_result;

int strq_set_pore_stats(strq_ctx* ctx, double tail_lo, double tail_hi, double model_min, double model_max);
int strq_target_add(strq_ctx* ctx, const float* prefix_ext, int64_t m_prefix, const float* suffix_ext,
                    int64_t m_suffix, int32_t trim_prefix, int32_t trim_suffix, int32_t samples,
                    int32_t hmm_model_id, int32_t count_bias, int32_t* target_id);
/* Base-modification pass of a target (repeatModHMM, STRique.py:447-500,605-609): the dual
 * (unmodified | modified) repeat-unit model uploaded with strq_model_create and its emission range
 * min(model_mins), max(model_maxs).  Reads of such a target also get a modification pattern. */
int strq_target_set_mod(strq_ctx* ctx, int32_t target_id, int32_t mod_model_id, double mod_min, double mod_max);
/* Modification patterns of the last batch: pattern of read i is pool[off[i] .. off[i+1]) ('0'/'1'
 * per repeat unit, "-" when there is none).  pool may be NULL to query the total size in off[n]. */
int strq_batch_fetch_mod(strq_ctx* ctx, char* pool, int64_t pool_cap, int64_t* off);
int strq_detect_batch(strq_ctx* ctx, int64_t n_reads, const void* signals, int32_t dtype,
                      const int64_t* offsets, const int32_t* target_id, const double* host_stats,
                      strq_result* out);
/* The same with one buffer per read (reads[i]: lengths[i] samples of `dtype`) instead of one concatenated buffer:
 * a caller that holds a list of arrays -- repeatCounter.detect_batch, the `count` command -- does not have to copy
 * them together first. */
int strq_detect_batch_reads(strq_ctx* ctx, int64_t n_reads, const void* const* reads, const int64_t* lengths, int32_t dtype,
                            const int32_t* target_id, const double* host_stats, strq_result* out);
/* The same in three steps, so that a caller can keep the signals resident in HBM and time (or
 * repeat) the device work alone: upload = host -> HBM copy, run = all kernels, fetch = results. */
int strq_batch_upload(strq_ctx* ctx, int64_t n_reads, const void* signals, int32_t dtype,
                      const int64_t* offsets, const int32_t* target_id, const double* host_stats);
/* strq_batch_upload in parts, so that a caller never holds more than one part in host memory: the first call (first_read = 0)
 * sizes the resident batch for total_reads reads / total_samples int16 samples (a hint: the buffer grows when the parts hold
 * more), every call uploads n_reads reads (signals +
 * offsets[0 .. n_reads], offsets relative to `signals`) behind the ones before it (first_read = reads uploaded so far).  Reads not
 * yet uploaded are empty.  int16 only (dtype 0).  bench.py stages its resident batches with it (synthesise -> upload -> free). */
int strq_batch_upload_part(strq_ctx* ctx, int64_t total_reads, int64_t total_samples, int64_t first_read, int64_t n_reads,
                           const void* signals, int32_t dtype, const int64_t* offsets, const int32_t* target_id);
int strq_batch_run(strq_ctx* ctx);
/* Only reads [first, last) of the uploaded batch (several batches kept resident side by side: a benchmark that
 * times a different one every step); strq_batch_fetch still returns the rows of the whole upload.
 * The run calls work in sub-batches (16 reads per CU) and keep TWO of them in flight: the HMM Viterbi launches of a sub-batch
 * run on a second stream under the conditioning and flank alignments of the next (the reference's workers are independent in the
 * same way, scripts/STRique.py:743-746), and its rows are taken one sub-batch late.  A run call therefore returns with the
 * Viterbi launches of its LAST sub-batch still queued; every call that hands out rows waits for what it needs:
 * strq_batch_fetch / strq_batch_fetch_mod / strq_detect_batch* for everything, strq_batch_fetch_range for the sub-batches that
 * hold reads of [first, last) only -- so `run(k + 1); fetch_range(k)` never waits for k + 1.  The rows are the same whatever
 * the order (STRQ_SERIAL=1: everything on one stream, rows before the run call returns, as up to ABI 11). */
int strq_batch_run_range(strq_ctx* ctx, int64_t first, int64_t last);
int strq_batch_fetch(strq_ctx* ctx, strq_result* out);
int strq_batch_fetch_range(strq_ctx* ctx, int64_t first, int64_t last, strq_result* out);
/* A host-side helper, not on the path of strq_detect_batch (which takes these on the GPU): the statistics of float64 reads
 * with numpy's arithmetic (no context, no device): out[6 * i ..] = median, MAD, c1, h1 of
 * medfilt(read i, 3) and c1, h1 of read i itself (0, 1 unless want_raw) -- what numpy's median / mean / percentile
 * give repeatCounter.detect (STRique.py:590-597). */
int strq_host_stats(const double* signals, const int64_t* offsets, int64_t n_reads, int32_t want_raw, double* out);
/* Host-side helper of the fast5 reader (strique_amd/vbz.py): the variable-byte layer of a VBZ chunk -- n integers of
 * isize (2 | 4) bytes from key_bits (2: StreamVByte, 1: the 16-bit variant) keys + data, optionally zig-zag coded
 * differences.  Returns the stream bytes consumed (the caller checks it against the stream length) or -1. */
int64_t strq_svb_decode(const uint8_t* stream, int64_t stream_len, int64_t n, int32_t key_bits, int32_t zigzag,
                        int32_t isize, void* out);
/* Host-side helper of the fast5 reader (strique_amd/fast5.py): n_chunks deflate-compressed chunks of a 1-D chunked HDF5
 * dataset, chunk k at file offset addr[k] (csize[k] stored bytes, first element elem_off[k]), inflated from the mapped file
 * `base` into out[n_total] (elements of elem_size bytes; shuffle != 0: undo the HDF5 shuffle filter).  What h5py does
 * for the reference's get_raw (STRique_lib/fast5Index.py:220-233).  Returns 0, -1 on a bad argument, -(k + 2) when
 * chunk k is out of bounds or does not inflate. */
int64_t strq_inflate_chunks(const uint8_t* base, int64_t base_len, int64_t n_chunks, const int64_t* addr,
                            const int32_t* csize, const int64_t* elem_off, int32_t elem_size, int32_t shuffle,
                            int64_t chunk_elems, int64_t n_total, void* out);
/* Host-side helper of the fast5 reader: resolve the 1-D dataset `path` (components separated by '/') below the old-style group
 * whose version-1 object header sits at `group_ohdr` of the mapped file, and list its chunks -- what h5py does when the reference
 * opens /read_<id>/Raw/Signal (STRique_lib/fast5Index.py:76-84,220-233).  meta = {elements, element size, 0 unsigned | 1 signed |
 * 2 float, layout 1 contiguous | 2 chunked, data address (contiguous) or chunk B-tree address, elements per chunk, filters
 * (bit 0 deflate, bit 1 shuffle before it, bit 2 VBZ alone), 1 when the chunks tile the dataset -- every element is then written by
 * the decoder, then the VBZ client data {version, integer size, zig-zag flag, zstd level}, four reserved zeros}.  Returns the number of chunks written to chunk_addr / chunk_size / chunk_off (0 for
 * a contiguous dataset), STRQ_H5_MORE_CHUNKS when max_chunks is too small, STRQ_H5_UNHANDLED for anything else -- a structure this
 * helper does not cover or a malformed file: the caller then takes its general (Python) path, which also reports errors. */
#define STRQ_H5_UNHANDLED (-100)
#define STRQ_H5_MORE_CHUNKS (-101)
int64_t strq_h5_locate(const uint8_t* base, int64_t base_len, int64_t group_ohdr, const char* path, int64_t meta[16],
                       int64_t* chunk_addr, int32_t* chunk_size, int64_t* chunk_off, int64_t max_chunks);
/* The same for n_ds datasets in one call (one reader task of the `count` command): dataset i owns the chunks
 * [chunk_first[i], chunk_first[i + 1]) of addr / csize / elem_off and has its own mapped file base[i].  status[i] receives what
 * strq_inflate_chunks would return for it; the return value is the number of datasets that failed (-1: bad argument). */
int64_t strq_inflate_many(int64_t n_ds, const uint8_t* const* base, const int64_t* base_len, const int64_t* chunk_first,
                          const int64_t* addr, const int32_t* csize, const int64_t* elem_off, const int32_t* elem_size,
                          const int32_t* shuffle, const int64_t* chunk_elems, const int64_t* n_total, void* const* out,
                          int64_t* status);
/* The same contract for chunks behind the VBZ filter (HDF5 filter 32020: Oxford Nanopore's vbz_compression, what MinKNOW writes;
 * strique_amd/vbz.py has the format -- unpinned: no VBZ file or plugin exists in the image or the reference tree) with client data
 * {version, integer size, zig-zag flag, zstd level}: zstd through the system's libzstd (dlopen), then strq_svb_decode.  -1 also for
 * a combination this helper does not decode: the caller's general path then does, or says why not. */
int64_t strq_vbz_chunks(const uint8_t* base, int64_t base_len, int64_t n_chunks, const int64_t* addr, const int32_t* csize,
                        const int64_t* elem_off, int32_t elem_size, int32_t version, int32_t isize, int32_t zigzag, int32_t level,
                        int64_t chunk_elems, int64_t n_total, void* out);
/* Diagnostics of the two calls above: out[0] = nanoseconds spent inside the inflate itself (all threads together), out[1] = zlib
 * streams inflated, out[2] = bytes produced, since the last reset; reset != 0 clears the counters. */
void strq_inflate_stats(int64_t out[3], int32_t reset);
/* 1 when libdeflate (dlopen of libdeflate.so.0) decodes the zlib streams of strq_inflate_chunks in this process, 0 when
 * zlib does (library absent, or STRQ_NO_LIBDEFLATE=1).  Same bytes either way. */
int strq_inflate_backend(void);
/* Test hook: conditioning outputs (8-bit morphology levels, their 256 float32 values, and
 * {median, MAD, c1/h1 of the filtered, morphology and raw signal, h2, c2}) of read `read` of the
 * last sub-batch processed by strq_batch_run. */
int strq_debug_conditioning(strq_ctx* ctx, int64_t read, uint8_t* levels, int64_t n, float* level_val,
                            double* scalars10);
/* Test hook, host only (no context, no device): the tables strq_model_set_positions would upload for this model --
 * out_lp[31 * 64] (log-probability of every column of the layout per lane, -inf where a lane has no such edge),
 * out_own[6 * 64] (state of every slot and lane, -1 if none, -2 for a virtual relay state), out_meta[10] = {slot, lane of
 * the two broadcast sources, of start and of end, then the low and high word of the lane mask of relayed hub states} -- the
 * caller provides TEN int32.  STRQ_ERR_UNSUPPORTED and the reason in `why` when the model is no profile chain.
 * tests/test_g2_layout.py drives a plain restatement of the kernel's time step with them. */
int strq_debug_g2_layout(int32_t n_states, int32_t silent_start, int32_t start, int32_t end,
                         const int32_t* in_ptr, const int32_t* in_src, const double* in_logp,
                         const int32_t* emis_kind, const int32_t* count_inc, const int32_t* kind, const int32_t* pos,
                         double* out_lp, int32_t* out_own, int32_t* out_meta, char* why, int32_t why_len);

/* Test hook, host only (no context, no device): the integer frame the upper-bound screen would run in for these alignment
 * parameters (open_h, ext_h, open_v, ext_v, dist_offset, dist_min; src/align_raw.h:84-103) and reads of up to `max_n` samples.
 * Returns 1 and out[6] = {scale, -ext_h * scale, -ext_v * scale, what is added to every table entry, float32 slack * scale,
 * columns below which candidate chunks merge}, or 0 when the parameters allow no screen (then the float32 DP runs over whole reads). */
int strq_debug_screen_plan(const float params[6], int32_t samples, int32_t max_n, int32_t out[6]);

/* Kernel timing of the last batched call, milliseconds (HIP events on the library's stream):
 * [0] table build  [1] forward DP  [2] trace pass  [3] total  [4] table entries re-evaluated on
 * the host  [5] conditioning  [6] Viterbi  [7] number of forward-DP kernel launches.  */
int strq_last_timing(const strq_ctx* ctx, float ms[8]);
/* Work counters of the last batched call (what the benchmark's roofline is priced with):
 * [0] forward-DP wave-steps (one step = two DP columns of every flank row, summed over all waves)
 * [1] DP columns computed, including the columns column segments recompute   [2] alignments
 * [3] waves per alignment (column segments) of the last forward launch   [4] score tables per CU
 * [5] 1 = 24-bit tables, 0 = float32   [6] rows per lane   [7] Viterbi time steps. */
int strq_last_counters(const strq_ctx* ctx, double out[8]);
/* Launch geometry of the last forward-DP launch of the last batched call (which kernel instance ran; the parity
 * tests assert it, the benchmark names the kernel of its roofline with it):
 * [0] waves per alignment (column segments)   [1] score tables per CU   [2] WPE template argument of
 * align_forward_seg_kernel<R, S, PK, SEG, WPE> (0: the one-wave kernels ran)   [3] rows per lane R
 * [4] 1 = 24-bit tables   [5] overlap (columns) the pieces were cut with first   [6] the worst-case overlap
 * [7] forward launch groups of the last sub-batch. */
int strq_last_geometry(const strq_ctx* ctx, int32_t out[8]);
/* Viterbi launches of the flanked-model decode (scripts/STRique.py:603) of the last sub-batch of the last batched call:
 * [0] launches   [1] of them on the register-resident kernel (viterbi_g2_kernel: one launch serves repeat profiles of
 * either parity)   [2] on the lane-layout kernels   [3] on the general kernel (viterbi_csr_kernel). */
int strq_last_viterbi_launches(const strq_ctx* ctx, int32_t out[4]);
/* Flank alignments of the last batched call whose first forward round -- column segments cut with the short, adaptive
 * overlap, or the screen's windows -- did not reach the score that certifies it, and which therefore ran the second round over their
 * whole read with the worst-case overlap: [0] such alignments, [1] all alignments (two per read).  What a read that does not contain
 * its flank costs.  (Alignments the coarse screen's second look resolved with more windows are not in [0]: strq_last_screen_mode [5].) */
int strq_last_second_round(const strq_ctx* ctx, int64_t out[2]);
/* The upper-bound screen of the last batched call (csrc/screen_kernels.hip: an integer DP over the whole read whose last-row
 * values bound the float32 ones of src/align_raw.h:106-158 from above, so that the exact DP only runs over the column windows
 * that can hold the optimum): [0] ms in align_screen_kernel   [1] alignments screened   [2] of them with windows
 * [3] without (their whole read ran)   [4] columns inside the windows   [5] screen wave-steps (one step = two columns of
 * every flank row)   [6] scale (scores are rounded up to multiples of 1 / scale)   [7] candidate chunks of 128 columns. */
int strq_last_screen(const strq_ctx* ctx, double out[8]);
/* Which screen the last sub-batch of the last batched call ran: [0] 0 none, 1 the fine screen (align_screen_kernel: one DP row per flank
 * row, bound within m / scale of the exact last row), 2 the coarse one (align_screen3_kernel by default: three flank rows per DP row, both flanks
 * of a read per wave, candidates taken with a margin)   [1] / [2] sub-batches for which the coarse / the fine screen stays paused
 * (it did not pay on the last one it ran on)   [3] the coarse screen's candidate margin in score units (x 1.3 / 1.75 at three / six
 * rows per DP row)   [4] flank rows per DP row of the screen that ran (3: align_screen3_kernel, 1: align_screen1_kernel, 0: the one-flank
 * kernel)   [5] alignments of the last batched call that missed the first look's certificate and were resolved by the coarse screen's second look
 * (or belonged to an attempt that started over with the fine screen)   [6..7] 0. */
int strq_last_screen_mode(const strq_ctx* ctx, int32_t out[8]);
/* The two sub-batches in flight, since the start of the last run call: [0] ms of HMM Viterbi launches whose rows were taken
 * [1] of them, ms that lay under the screen kernel of the sub-batch that followed   [2] ... under its whole alignment stage
 * (screen, exact pass, trace)   [3] sub-batches counted in [1], [2] (those whose rows were taken by the following sub-batch's run
 * call; a sub-batch nobody followed runs its Viterbi launches alone). */
int strq_last_overlap(const strq_ctx* ctx, double out[4]);

#ifdef __cplusplus
}
#endif
#endif
