/*
 * issl_hip.h -- C ABI of libissl_hip.so, the MI355X (gfx950) ISSL off-target scorer.
 *
 * Drop-in scope: the `isslScoreOfftargets` step of Crackling (reference paths relative to
 * /root/reference).  The reference has no in-process API for this step -- its boundary is the
 * process `isslScoreOfftargets <issl> <query> <maxDist> <threshold> <method>` launched from
 * src/crackling/Crackling.py:767-778 -- so every entry point below names the block of
 * src/ISSL/isslScoreOfftargets.cpp (or isslCreateIndex.cpp) whose work it takes over.
 * bin/isslScoreOfftargets (crackling_amd/csrc/cli_score.cpp) is the shipped caller.
 *
 * Conventions: plain C types only; every function returns 0 on success or a negative
 * ISSL_E_* code, never throws and never calls exit(); issl_last_error() holds the message of the
 * last failure on the calling thread.  One issl_index is used by one thread at a time.
 * Scoring needs a HIP device: there is NO CPU fallback, calls fail with ISSL_E_DEVICE.
 */
#ifndef ISSL_HIP_H
#define ISSL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISSL_ABI_VERSION 6

enum {
    ISSL_OK = 0,
    ISSL_E_ARG = -1,      /* bad argument */
    ISSL_E_IO = -2,       /* file cannot be opened / read / written */
    ISSL_E_FORMAT = -3,   /* .issl or query file malformed (reference: "Error reading index", exit 1) */
    ISSL_E_UNSUPPORTED = -4, /* geometry the kernels do not implement */
    ISSL_E_DEVICE = -5,   /* HIP error or no device */
    ISSL_E_NOMEM = -6,
    ISSL_E_STATE = -7,    /* call order (e.g. score before upload) */
    ISSL_E_RETRY = -8     /* issl_score_finish: a batch ran out of scratch space (buffers grown) or needs the whole pipeline: enqueue it again */
};

/* Score methods, isslScoreOfftargets.cpp:44,121-143. */
enum {
    ISSL_METHOD_UNKNOWN = 0, /* neither score computed; the CLI prints -1 -1 */
    ISSL_METHOD_MIT = 1,
    ISSL_METHOD_CFD = 2,
    ISSL_METHOD_AND = 3,
    ISSL_METHOD_OR = 4,
    ISSL_METHOD_AVG = 5
};

typedef struct issl_index issl_index; /* opaque */

/* Header of an .issl file, isslScoreOfftargets.cpp:162-174 / isslCreateIndex.cpp:257-263. */
typedef struct {
    uint64_t n_sites;     /* offtargetsCount: distinct sites */
    uint64_t seq_len;     /* seqLength */
    uint64_t n_lines;     /* seqCount: input lines including duplicates */
    uint64_t slice_width; /* bits per slice */
    uint64_t n_slices;    /* sliceCount */
    uint64_t n_scores;    /* scoresCount: {mask, local MIT score} pairs in the file */
} issl_header;

/* One scored off-target (the reference keeps these only implicitly, :382-463). */
typedef struct {
    uint32_t guide; /* index into the guide batch */
    uint32_t slice; /* slice whose bucket produced it (first matching slice) */
    uint32_t pos;   /* position j inside that bucket, :344 */
    uint32_t id;    /* site id = low 32 bits of the bucket entry, :347 */
    uint32_t dist;  /* mismatches, :380 */
    uint32_t occ;   /* occurrences = high 32 bits of the entry, :348 */
} issl_hit;

/* Timings and counters of the last issl_score* call on an index (milliseconds).  ms_scan is always measured (mean over
 * the batches since the last finish) by the scan kernel itself: first workgroup in to last workgroup out on the 100 MHz
 * constant clock (s_memrealtime), i.e. the launch's own duration; the other stage times (GPU events) and ms_total are filled by the
 * synchronous entry points, and by issl_score_device_async only when the stage_timing option is set (every event
 * record costs ~4 us of stream time, which back-to-back batches should not pay). */
typedef struct {
    uint64_t n_guides;
    uint64_t candidates;    /* (guide, candidate) comparisons the scan kernel COUNTED while making them: real
                               candidates of every tile it fetched x real guides it ran past it.  Full scan: equals
                               reference_comparisons.  Pruned scan: far fewer (only the successor-byte groups of a
                               bucket that can hold a hit are fetched), and >= planned_comparisons because a group's
                               first and last tile also hold neighbours' candidates */
    uint64_t hits;          /* candidates within max_dist, first matching slice only */
    uint64_t scan_tiles;    /* candidate tiles x guide groups processed by the scan kernel */
    double ms_bin;          /* guide binning kernels */
    double ms_scan;         /* XOR/popcount scan kernel (the roofline kernel) */
    double ms_verify;       /* exact re-test of the candidates the scan noted, first-matching-slice rule */
    double ms_group;        /* hit grouping (count/scan/scatter) */
    double ms_replay;       /* ordered MIT/CFD accumulation */
    double ms_total;        /* first kernel to last kernel */
    uint64_t scan_launches; /* >1 when a hit buffer had to grow and the scan was repeated */
    uint64_t raw_records;   /* upper bound of candidates noted by the scan (chunks handed out x chunk size) */
    uint64_t n_batches;     /* batches covered by these statistics (ms_scan is their mean) */
    uint64_t planned_comparisons; /* what the planning kernel expected the scan to compare: guides x lengths of the
                                     buckets (full scan) or successor-byte groups (pruned scan) they visit */
    uint64_t reference_comparisons; /* sum over guides of their five bucket lengths = iterations of the reference's
                                       loop :344 without early exit (host equivalent: issl_count_candidates) */
    uint64_t pruned;              /* 0: full scan; 1 / 2 / 3: pruned scan over the successor-byte groups equal to / within
                                     one / within two mismatches of the guide's own (max_dist <= 2 / <= 4 / = 5, sorted image) */
    double ms_scan_events;        /* the scan launches by the HIP event pair recorded around them on their stream (mean over
                                     the batches that recorded one: scan_events option); equals ms_scan for batches on one
                                     lane, includes the wait for wave slots when a second lane shares the device (lanes) */
} issl_stats;

const char *issl_last_error(void);
int issl_abi_version(void);

/* ---- index: host side (A1, isslScoreOfftargets.cpp:152-270) ------------------------------ */

/* Map and validate an .issl file.  Errors the reference reports (:164-167,201-204,223-226,237-240)
 * and the ones it leaves undefined (missing file, truncated sections, ids out of range). */
int issl_index_open(const char *path, issl_index **out);

/* Same from a memory image of the file (copied). */
int issl_index_from_memory(const void *image, size_t len, issl_index **out);

/* isslCreateIndex.cpp:132-289 counterpart: build an index from the text of a SORTED site list,
 * n_lines lines of seq_len characters + '\n'.  issl_index_write() then emits the same bytes as
 * the reference builder. */
int issl_index_build_from_text(const char *text, size_t n_lines, size_t seq_len,
                               size_t slice_width, issl_index **out);

/* Build from already packed, de-duplicated signatures in id order (isslCreateIndex.cpp:199-200
 * state after the counting loop): sigs[i] occurs occ[i] times; n_lines = sum(occ). */
int issl_index_build_from_sites(const uint64_t *sigs, const uint32_t *occ, size_t n_sites,
                                size_t n_lines, size_t seq_len, size_t slice_width,
                                issl_index **out);

/* Same inputs, but the slice lists (isslCreateIndex.cpp:218-234) are built on `device`: the signatures are
 * copied into the HBM image and one stable radix pass per slice writes the lists next to them.  The result is
 * uploaded and ready to score; the host keeps only the geometry, the MIT table and the bucket sizes (12 B/site of
 * input instead of 48 B/site of host arrays), so indexes up to the format's 2^32-1 sites fit a 288 GB GPU.
 * issl_index_write() streams the arrays back out of the image (same bytes as the host builder);
 * issl_index_upload() to another device is refused -- replicate with issl_index_image/attach_image.
 * 20 bp; slices of 8, 4 or 2 bits (the scorer's geometries). */
int issl_index_build_on_device(const uint64_t *sigs, const uint32_t *occ, size_t n_sites,
                               size_t n_lines, size_t seq_len, size_t slice_width, int device,
                               issl_index **out);
/* The same with layout options for the image it makes, as issl_index_set_option would set them on a handle before
 * issl_index_upload: `options` = "key=value,key=value" (e.g. "compact=1,host_cold=1") or NULL.  (ABI 4) */
int issl_index_build_on_device_opt(const uint64_t *sigs, const uint32_t *occ, size_t n_sites,
                                   size_t n_lines, size_t seq_len, size_t slice_width, int device,
                                   const char *options, issl_index **out);

/* The same for a site table that is ALREADY in the memory of `device` (d_sigs, d_occ: device pointers; e.g. the sorted,
 * de-duplicated keys issl_extract_* leaves there, or sites generated on the GPU): bucket lengths are counted on the
 * device, nothing of the size of the index ever exists in host memory.  The inputs are copied into the image and may be
 * freed when the call returns.  isslCreateIndex.cpp:199-234 from its state after the counting loop on.  (ABI 5)
 * Checked on the device: no signature carries bits above 2 * seq_len (ISSL_E_ARG).  NOT checked, as in the host-side
 * builders and in the reference (which collapses only CONSECUTIVE equal lines, isslCreateIndex.cpp:189-193): that the
 * signatures are distinct -- a signature listed twice is two sites, found and scored twice, exactly as the reference scores
 * an index built from unsorted text. */
int issl_index_build_from_device_sites(const uint64_t *d_sigs, const uint32_t *d_occ, size_t n_sites,
                                       size_t n_lines, size_t seq_len, size_t slice_width, int device,
                                       const char *options, issl_index **out);

/* Write the .issl bytes (isslCreateIndex.cpp:256-289). */
int issl_index_write(const issl_index *idx, const char *path);

int issl_index_header(const issl_index *idx, issl_header *out);

/* Bucket lengths, slice-major (isslScoreOfftargets.cpp:221-226): n_slices << slice_width values. */
int issl_index_bucket_sizes(const issl_index *idx, uint64_t *out, size_t n);

int issl_index_close(issl_index *idx);

/* ---- index: HBM image ----------------------------------------------------------------- */

/* Bytes of the device image (sites + bucket entries + packed scan stream + tables). */
int issl_index_device_bytes(const issl_index *idx, size_t *out);

/* Free and total memory of `device` in bytes (hipMemGetInfo): what a resident server weighs an upload against.  (ABI 4) */
int issl_device_memory(int device, size_t *free_bytes, size_t *total_bytes);

/* Allocate the image on `device` (hipMalloc), copy and transform.  Owned by the index. */
int issl_index_upload(issl_index *idx, int device);

/* Same into caller-owned device memory (e.g. a torch uint8 tensor) of at least
 * issl_index_device_bytes() bytes, 256-byte aligned.  The caller keeps it alive until close. */
int issl_index_upload_into(issl_index *idx, int device, void *dev_buf, size_t bytes);

/* Adopt an image that some other rank produced and that arrived in `dev_buf` through an RCCL
 * broadcast: no file is needed on this rank.  Creates a new index handle. */
int issl_index_attach_image(int device, void *dev_buf, size_t bytes, issl_index **out);

/* Device pointer/size of the current image (for the broadcast on the producing rank). */
int issl_index_image(const issl_index *idx, void **dev_ptr, size_t *bytes);

/* Copy the image into caller-owned device memory of the same device (256-byte aligned, >= issl_index_image bytes):
 * how an index that was built on the device gets into the tensor a framework broadcasts. */
int issl_index_copy_image_to(const issl_index *idx, void *dev_dst, size_t bytes);

/* Image layouts, and indexes larger than the free HBM (BASELINE configs[4]; the format's 32-bit ids,
 * isslScoreOfftargets.cpp:347, allow 4.29 G sites).  Only the scan stream (20 B/site) is read by the scan; everything else
 * is touched for the ~2e-5 of the comparisons that come within max_dist.  issl_index_upload / issl_index_build_on_device
 * try, in this order, until one fits the free HBM:
 *   sorted        every bucket ordered by the byte of the next slice (what the pruned scan needs) + 16-byte stream records,
 *                 site table, counts, slice lists: 152 B/site
 *   compact       the same order with 4-byte site ids per stream position: 92 B/site, or 52 B/site WITHOUT the slice lists
 *                 (ABI 5; the automatic fallback): scoring never reads them, and on a sorted layout they are a function of
 *                 site table and counts (every list ascends by site id, isslCreateIndex.cpp:218-234), which
 *                 issl_index_write and issl_dump_hits redo on the device when asked.  An index at the format's limit of
 *                 4.29 G sites takes 223 GB of a 288 GB GPU (3 G lines: 152 GB, measured), nothing in host memory, and
 *                 the image still moves as one broadcast.  While a sorted image is built it needs 8 B per site beyond its
 *                 own size (keys and slice list of the slice being ordered borrow the image's scan section, which is packed
 *                 last): 60 B/site = 258 GB at the format's limit from a file or host arrays (+ a 2 - 9 GB working reserve);
 *                 issl_index_build_from_device_sites has the caller's 12 B/site beside it (~3.8 G sites on 288 GB).  On request (host_cold=1) the lists are kept in pinned,
 *                 mapped HOST memory instead (40 B/site there)
 *   list order    (indexes whose lists do not ascend by site id, or whose five lists disagree about a site's count: no
 *                 builder writes such, the reference does not care; no pruned scan then) with / without in-list
 *                 signatures 108 / 68 B/site, or with site table and lists in host memory 25 B/site (the kernels rebuild
 *                 signatures from the scan stream and read host memory only for counts >= 255 and for issl_dump_hits)
 * Indexes with ten 4-bit or twenty 2-bit slices take every layout but the last (the sorted ones order a bucket by the byte
 * of the next two / four slices; per site the slice lists then cost 80 / 160 B instead of 40).
 * Options sorted_layout / compact / inline_sigs / host_cold force a choice (or the upload fails); results are identical
 * in every layout.  An index with a site in a bucket its signature does not select, or twice in one slice, is refused
 * (ISSL_E_FORMAT).  issl_index_cold() returns the host buffer of an image with host-resident sections (NULL / 0 when
 * everything is in HBM); another device of the same process adopts a copy of the hot image plus the SAME host buffer
 * with issl_index_attach_image_cold (issl_node does this).  Such images cannot be attached in another process. */
int issl_index_cold(const issl_index *idx, void **host_ptr, size_t *bytes);
int issl_index_attach_image_cold(int device, void *dev_buf, size_t bytes, void *cold_host, size_t cold_bytes,
                                 issl_index **out);

/* Tuning knobs.  Every knob has an environment variable that is read ONCE, when the handle is created (open / build /
 * attach), never inside a scoring call; afterwards this call changes it (no batches may be in flight).  Keys (env):
 *   scan_blocks (ISSL_SCAN_BLOCKS) workgroups of the scan launch      item_guides (ISSL_ITEM_GUIDES) guides per scan item
 *   scan_generic (ISSL_SCAN_GENERIC) 0|1 runtime-threshold scan       stage_timing (ISSL_STAGE_TIMING) 0|1 events at every stage
 *   scan_events (ISSL_SCAN_EVENTS) 0|1|2: the HIP event pair around the scan of an asynchronous batch (issl_stats::ms_scan_events):
 *     1 every batch, 2 (default) the first batch after a finish, 0 never -- an event record is ~5 us of stream time, and
 *     ms_scan (the kernel's own clock stamps) times every launch without them
 *   raw_chunks (ISSL_RAW_CHUNKS) initial raw-record buffer
 *   sorted_layout (ISSL_SORTED_LAYOUT), compact (ISSL_COMPACT), inline_sigs (ISSL_INLINE_SIGS), host_cold
 *     (ISSL_FORCE_HOST_COLD), keep_lists (ISSL_KEEP_LISTS), each -1|0|1: image layout, read at upload (see above; -1 =
 *     automatic).  compact=1 with host_cold=1: the compact sorted image with its slice lists in host memory; host_cold=1
 *     alone: the list-order image with site table and lists in host memory; keep_lists=0: the compact sorted image
 *     without slice lists (52 B/site), keep_lists=1: never drop them
 *   tail_shapes (ISSL_TAIL_SHAPES) 0|1 (default 1): the short last unit of a successor-byte group runs 2 / 4 guides per
 *     pass on 16 / 8 candidates per lane
 *   hit_slots (ISSL_HIT_SLOTS) 0|1|2 (default 1): the first 512 hits of every guide go straight from the exact test to a
 *     32-byte record of their own (1.6 GB of scratch per 100 000 guides of a batch; batches beyond 512 k guides, a
 *     device short of memory and issl_dump_hits go without; a handle whose batches show many guides beyond 512 hits
 *     widens its slots to 2048 by itself); 0: every hit passes through the grouping pass; 2: slots for 2048 hits per
 *     guide from the first batch on (tests and A/B: the results do not depend on the width)
 *   lean_tail (ISSL_LEAN_TAIL) 0|1 (default 1): a handle whose finished batches met no guide beyond its hit slots
 *     enqueues the next ones without the grouping pass and the three many-hit replays (five dependent launches that
 *     would find nothing to do: 25 us of every batch); a batch that does meet such a guide is run again in full
 *   small_bin (ISSL_SMALL_BIN) 0|1 (default 1): batches of up to 512 (guide, slice) pairs (102 guides of five slices), max_dist
 *     <= 4, sorted image, are binned in two launches instead of seven -- every placement a group of its own (a matter of
 *     latency only: 64 guides against 300 M sites 0.110 -> 0.079 ms); fine_items (ISSL_FINE_ITEMS): initial capacity of the
 *     pruned plan's item list instead of the size derived from the index (tests of the list's two overflow paths)
 *   expect_guides (ISSL_EXPECT_GUIDES) n: a batch of about n guides follows the upload at once (the one-shot scorer knows its
 *     page): the scoring workspace's streams, events and small buffers are set up on a thread of their own beside the upload --
 *     20 ms less in front of the first kernel; 0 (default): nothing is prepared
 *   upload_chunk_kib, upload_ring_min_kib, upload_threads (ISSL_UPLOAD_CHUNK_KIB, ISSL_UPLOAD_RING_MIN_KIB,
 *     ISSL_UPLOAD_THREADS): the ring of pinned chunks a file-mapped index is uploaded through (eight threads pread the
 *     file into two slots each, every slot leaves with its own asynchronous copy: the PCIe link's rate, where hipMemcpy
 *     from the fresh mapping moves a fifth of it; the sections are queued one behind the other, every reader pins its
 *     slots when it first needs them, and the ring goes back on a thread of its own after the handle's first scoring
 *     call -- or at issl_index_close): KiB per slot (default 16384), the section size from which the ring is used
 *     (default 65536), readers (1..32, default 8)
 *   scan_threads (ISSL_SCAN_THREADS) 64..1024: threads per scan workgroup (default 1024 = 8 waves per SIMD; an occupancy
 *     experiment)
 *   prune (ISSL_PRUNE) -1|0|1: scan only the successor-byte groups of a bucket that can hold a site within max_dist (13
 *     of 256 for max_dist <= 4, 1 of 256 for <= 2, 67 of 256 for max_dist 5; same hits and scores as the reference's scan
 *     of the whole bucket, isslScoreOfftargets.cpp:344): -1 = a planning kernel decides per batch from the two plans'
 *     estimated times, 0 = never, 1 = whenever the image is sorted and max_dist <= 5
 *   lanes (ISSL_LANES) 1|2|3: workspaces that the batches of issl_score_device_async alternate between (default 1).  With
 *     2 the batches form a software pipeline: scans one after the other, verify / group / replay of a batch on a
 *     high-priority stream beside the next batch's scan; with 3 only the BINNING of a batch (its short, latency-bound
 *     launches) runs beside the batch before it, the scan waits for that batch's replay.  Outputs of two consecutive
 *     batches must be different buffers with either
 *   scan_stamps (ISSL_SCAN_STAMPS) file for per-wave clocks (diagnostics) */
int issl_index_set_option(issl_index *idx, const char *key, const char *value);
/* Current value of an integer knob; also the read-only keys is_sorted, is_compact, cold_on_host, cold_sections (0, 1 =
 * slice lists, 3 = lists + site table in host memory), lists_absent, has_inline_sigs and dense_mit (layout of the uploaded image, -1
 * before an upload). */
int issl_index_get_option(const issl_index *idx, const char *key, long long *value);

/* ---- guides (A2, isslScoreOfftargets.cpp:63-71,82-89,275-305) ---------------------------- */

/* 2-bit pack n guides laid out as lines of `stride` bytes (seq_len chars then anything). */
int issl_encode_guides(const char *text, size_t n, size_t seq_len, size_t stride, uint64_t *out);
/* out must hold seq_len+1 bytes. */
int issl_decode_guide(uint64_t sig, size_t seq_len, char *out);
/* Query file rules of :275-294; *out is malloc'd (free with issl_free).  Large files are read and packed by several
 * threads. */
int issl_read_query_file(const char *path, size_t seq_len, uint64_t **out, size_t *n);

/* ---- output text (A12, isslScoreOfftargets.cpp:514-527) ------------------------------------- */
/* The scorer's stdout for n guides, in input order: "<seq>\t<MIT>\t<CFD>\n" with both scores as printf("%f") prints
 * them and "-1" for a score `method` (ISSL_METHOD_*) does not ask for (:517-525; ISSL_METHOD_UNKNOWN: "-1\t-1").  The text
 * comes in *n_spans consecutive pieces (formatted by up to `threads` threads, 0 = automatic; a million lines are
 * otherwise as long as their scoring): write them out one after the other, release with issl_free_spans.  The "%f" is
 * the library's own exact formatter (the binary value rounded to six decimals, ties to even: digit for digit glibc's
 * output), snprintf for negative, huge and non-finite values.  Host arithmetic only.  issl_free_spans hands the buffers
 * back to the library, which keeps up to 512 MB of them for the next call (a resident scorer formats page after page of the
 * same size, and a fresh buffer costs a page fault per 4 KiB when it is first written).  (ABI 6) */
typedef struct issl_span {
    char *data;
    size_t len;
} issl_span;
int issl_format_scores(const uint64_t *guides, const double *mit, const double *cfd, size_t n, size_t seq_len,
                       int method, int threads, issl_span **spans, size_t *n_spans);
void issl_free_spans(issl_span *spans, size_t n_spans);
void issl_free(void *p);

/* ---- scoring (A3-A11, isslScoreOfftargets.cpp:307-511) ------------------------------------ */

int issl_method_from_string(const char *s);

/* Score n guides held in host memory; mit/cfd receive 10000/(100+sum) per guide (:505-506).
 * Blocking.  Both outputs are always written (the reference prints -1 for a method that was not
 * requested, :517-525 -- that is the caller's business). */
int issl_score(issl_index *idx, const uint64_t *guides, size_t n, int max_dist, double threshold,
               int method, double *mit, double *cfd);

/* Same with guides and outputs already in device memory of the index's device; asynchronous
 * on `stream` (a hipStream_t, may be NULL) except when the hit buffer must grow. */
int issl_score_device(issl_index *idx, const uint64_t *d_guides, size_t n, int max_dist,
                      double threshold, int method, double *d_mit, double *d_cfd, void *stream);

/* Enqueue one batch and return at once; any number of batches may be enqueued before issl_score_finish().
 * Batches run back to back on an internal stream of the index, independent of the caller's streams; use different
 * output buffers for batches whose results are consumed later.
 * `stream`: if not NULL the batch starts after the work enqueued on that stream so far (the producer of d_guides);
 * NULL means the inputs are ready now.  Results are valid after issl_score_finish(), or, on a stream, behind
 * issl_score_wait(idx, stream), which makes that stream wait for every batch enqueued so far (no host sync).
 * issl_score_finish() synchronises (internal streams and `stream`) and checks the batches: ISSL_E_RETRY means that a
 * batch needed more scratch space than was allocated (first large batch on an index; the buffers have been enlarged), or
 * that it was enqueued without the many-hit part of the pipeline (lean_tail option: a handle whose batches meet no guide
 * with more than 512 hits stops launching it) and did meet such a guide: the batches since the previous finish must be
 * enqueued again.  issl_last_stats() then describes the last
 * batch, with ms_scan averaged over all of them. */
int issl_score_device_async(issl_index *idx, const uint64_t *d_guides, size_t n, int max_dist,
                            double threshold, int method, double *d_mit, double *d_cfd, void *stream);
int issl_score_wait(issl_index *idx, void *stream);
int issl_score_finish(issl_index *idx, void *stream);

/* Parity helper: the scored off-targets of every guide in the reference's scoring order
 * (slice, then position in bucket), truncated by early exit exactly as :467-496.
 * *n_hits receives the total even when it exceeds cap. */
int issl_dump_hits(issl_index *idx, const uint64_t *guides, size_t n, int max_dist,
                   double threshold, int method, issl_hit *hits, size_t cap, size_t *n_hits);

int issl_last_stats(const issl_index *idx, issl_stats *out);

/* Sum over guides of the five bucket lengths (SURVEY 8d cross-check), host arithmetic only. */
int issl_count_candidates(const issl_index *idx, const uint64_t *guides, size_t n, uint64_t *out);

/* ---- caller-side thresholding (src/crackling/Crackling.py:780-835) ----------- */
/* What Crackling does with the scorer's stdout: the scores are read back from the "%f" text (6 decimals; -1 for a
 * score whose method was not requested, isslScoreOfftargets.cpp:517-525) and compared with the threshold under the
 * lower-cased, stripped method name (`mit`: MIT < t rejects; `cfd`; `and`: both below; `or`: either; `avg`: mean).
 * `method` is the string as configured: the scorer matches it exactly (:121-143) while the caller lower-cases it,
 * so "AND" prints -1/-1 and then rejects -- reproduced here.  accepted[i] = ISSL_VERDICT_REJECTED (0, CODE_REJECTED),
 * ISSL_VERDICT_ACCEPTED (1, CODE_ACCEPTED) or ISSL_VERDICT_NONE (the caller's if/elif chain matches no method and
 * leaves the guide untouched).  Host arithmetic only.  bin/isslScoreOfftargets writes "<20-mer>\t<0|1>\n" lines to
 * the file named by ISSL_VERDICTS when that variable is set (stdout is unchanged). */
enum { ISSL_VERDICT_REJECTED = 0, ISSL_VERDICT_ACCEPTED = 1, ISSL_VERDICT_NONE = 255 };
int issl_verdicts(const double *mit, const double *cfd, size_t n, double threshold, const char *method,
                  uint8_t *accepted);

/* ---- one process, several GPUs of the node ------------------------------------------------ */
/* The reference's outer loop is data-parallel over guides (isslScoreOfftargets.cpp:316-509 reads only the
 * index): a node replicates the HBM image on every listed device -- uploaded once on devices[0], then broadcast
 * with RCCL (ncclBroadcast over xGMI; peer copies when RCCL cannot be used, e.g. a device listed twice) -- and
 * scores every batch as a QUEUE OF CHUNKS (16 k - 256 k guides), one host thread per device taking the next chunk
 * when it has finished its last: the reference's static OpenMP split (:316) would leave the device that holds a
 * repeat-dense stretch of Crackling's genome-ordered guides working long after the others.  Scores land in input
 * order.  An image whose cold sections live in pinned host memory (issl_index_cold) is replicated hot part only, all
 * devices read the one host copy.  bin/isslScoreOfftargets uses a node when ISSL_DEVICES names several devices. */
typedef struct issl_node issl_node;

typedef struct {
    int n_devices;
    int used_rccl;         /* 1: image moved by ncclBroadcast, 0: hipMemcpyPeer */
    double ms_upload;      /* host -> devices[0] including the scan-stream transform */
    double ms_broadcast;   /* devices[0] -> all others */
    double ms_last_score;  /* wall time of the last issl_node_score call */
} issl_node_info;

/* idx has host arrays (opened from a file or built) or was built on devices[0] (issl_index_build_on_device); it stays
 * owned by the caller and must outlive the node.  devices may be NULL: then all visible devices are used. */
int issl_node_create(issl_index *idx, const int *devices, int n_devices, issl_node **out);
int issl_node_score(issl_node *node, const uint64_t *guides, size_t n, int max_dist, double threshold,
                    int method, double *mit, double *cfd);
int issl_node_get_info(const issl_node *node, issl_node_info *out);
/* Per device, for the last issl_node_score call: milliseconds spent scoring and guides scored (n >= n_devices). */
int issl_node_shard_times(const issl_node *node, double *busy_ms, uint64_t *guides, int n);
int issl_node_close(issl_node *node);

/* ---- off-target site extraction (the step before the index builder) ------------------------------ */
/* Counterpart of src/crackling/utils/extractOfftargets.py: every N20 site next to a PAM on either strand
 * (patterns of :23-24, overlapping matches, first 20 characters of the match, reverse-complemented for the
 * reverse pattern), sorted as text, duplicates kept, one per line.  Runs on `device`; no CPU fallback.
 * files[i]/lens[i]: the bytes of FASTA / multi-FASTA files.  *out_text is malloc'd (issl_free). */
int issl_extract_from_memory(const char *const *files, const size_t *lens, int n_files, int device,
                             char **out_text, size_t *out_len, uint64_t *n_sites);
/* Same from files on disk into `output_path` (what bin/extractOfftargets calls). */
int issl_extract_offtargets(const char *const *inputs, int n_inputs, const char *output_path, int device,
                            uint64_t *n_sites);

#ifdef __cplusplus
}
#endif
#endif /* ISSL_HIP_H */
