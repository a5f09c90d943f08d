/* fdcm.h -- C ABI of libfdcm_hip.so: the MI355X (gfx950) engine for OpenFDCM's hot path.
 *
 * The reference (Innoptech/OpenFDCM v0.10.0) has no C ABI or plug-in loader: its extension point
 * is a C++ type deriving from FeatureMapInstance (modules/matching/include/openfdcm/matching/
 * featuremap.h:11-15) with getFeatureSize / minmaxTranslation / evaluate specialised
 * (featuremap.h:27-52), driven by search<DefaultMatch> (modules/matching/src/matchstrategies/
 * defaultmatch.cpp:32-89) and optimize<BatchOptimize> (modules/matching/src/optimizestrategies/
 * batchoptimize.cpp:6-123), and exposed to Python by modules/python/src/matching.cpp.  These
 * entry points are what a binding for that path would call; INTEGRATION.md shows the stubs.
 *
 * Conventions: every function returns 0 on success and a negative FDCM_E* code on failure;
 * fdcm_last_error() returns a thread-local message.  No exception crosses the boundary.  Inputs
 * are caller-owned host buffers (plain pointers + counts) unless a parameter says "device".
 * Outputs allocated by the library are released with the matching *_free.  Calls block until the
 * result is complete and are safe to make with the Python GIL released; the one exception is
 * fdcm_featuremap_build / _rebuild, which return once the build is queued on the handle's HIP stream
 * (every later call on the handle that touches the volume -- search, slice, device_volume, timing,
 * free -- is ordered behind it or waits for it, so a kernel failure is reported by that call).  Handles are
 * thread-compatible (one caller at a time per handle).
 *
 * Line arrays are the reference's LineArray (math.h:66): 4 x N float32, column-major, i.e. N
 * consecutive records x1,y1,x2,y2.
 */
#ifndef FDCM_H
#define FDCM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FDCM_OK 0
#define FDCM_EINVAL (-1)   /* bad argument */
#define FDCM_EHIP (-2)     /* a HIP runtime call failed (no device, out of memory, launch error) */
#define FDCM_EINTERNAL (-3)

/* core::Distance, modules/core/include/openfdcm/core/imgproc.h:148 */
enum fdcm_distance { FDCM_L2 = 0, FDCM_L2_SQUARED = 1, FDCM_L1 = 2 };
/* optimiser strategies: defaultoptimize.cpp:6-93, batchoptimize.cpp:6-123, indulgentoptimize.cpp:6-102.
 * For FDCM_INDULGENT_OPTIMIZE the batch_size argument of the search calls is the number of passthroughs (it
 * does not change the result: a passed-through score is re-scored at the same translation). */
enum fdcm_optimizer { FDCM_DEFAULT_OPTIMIZE = 0, FDCM_BATCH_OPTIMIZE = 1, FDCM_INDULGENT_OPTIMIZE = 2 };
/* penalty strategies: defaultpenalty.cpp:29-58, exponentialpenalty.cpp:34-64 */
enum fdcm_penalty { FDCM_DEFAULT_PENALTY = 0, FDCM_EXPONENTIAL_PENALTY = 1 };

/* matching::Match, matchstrategy.h:35-44; transform is the 2x3 matrix in row-major order. */
typedef struct fdcm_match {
    int32_t tmpl_idx;
    float score;
    float transform[6];
} fdcm_match;

typedef struct fdcm_featuremap fdcm_featuremap; /* replaces Dt3Cpu, dt3cpu.h:46-63 */
typedef struct fdcm_templates fdcm_templates;   /* a std::vector<LineArray> resident in HBM */

typedef struct fdcm_featuremap_info {
    int64_t width, height; /* getFeatureSize(): Size(x = W, y = H), always square (dt3cpu.cpp:113-115) */
    int64_t depth;         /* number of orientation slices actually built (distinct keys) */
    float scene_translation[2]; /* getSceneTranslation() */
    int32_t distance;
    float dt3_coeff, padding;
} fdcm_featuremap_info;

/* Per-stage device times of the last build, in milliseconds (HIP events on the build stream). */
typedef struct fdcm_build_timing {
    float total_ms;     /* host preparation + the kernels' span on the device */
    float seeds_ms;     /* K0: rasterise scene lines into the seed bitmap (feature sizes above 4096 only: below, K1 draws the seeds itself) */
    float pass1_ms;     /* K1: 1-D distance along y */
    float pass2_ms;     /* K2: in-place lower-envelope pass along x (L2/L2^2) or L1 sweeps */
    float propagate_ms; /* K3: orientation propagation (+ sqrt for L2) */
    float integral_ms;  /* K4: directional line integral */
    float span_ms;      /* the kernels' span on the device alone (first to last event; 0 without events) */
} fdcm_build_timing;

typedef struct fdcm_search_timing {
    float total_ms;  /* the search's span on the device: kernels + download of the matches */
    float kernel_ms; /* candidate generation + optimisation + compaction on the device, from the event that precedes them on the
                      * handle's stream.  A handle that has the GPU to itself runs the search's preparation (scene upload,
                      * candidate pairs, work list) on a second stream beside a build that is still running: the part of it
                      * that overlaps the build is then not inside kernel_ms / total_ms */
    int64_t candidates;
    int64_t evaluations; /* translations scored by the reference rule (kept + rejected batches) */
} fdcm_search_timing;

const char* fdcm_last_error(void);
const char* fdcm_version(void);

int fdcm_device_count(int* count);
int fdcm_set_device(int device); /* device used by handles created afterwards on this thread */
int fdcm_get_device(int* device); /* the device fdcm_set_device selected on this thread (0 by default) */

/* ---- DT3 feature map: buildCpuFeaturemap<D>, dt3cpu.h:174-234; Python build_cpu_featuremap,
 *      modules/python/src/matching.cpp:116-130 ---- */
int fdcm_featuremap_build(const float* scene_lines, int64_t n_lines, int64_t depth, float dt3_coeff, float padding,
                          int distance, fdcm_featuremap** out);
/* Rebuild into an existing handle (same depth/distance parameters), reusing its HBM when the
 * feature size allows: the steady-state per-frame call. */
int fdcm_featuremap_rebuild(fdcm_featuremap* fm, const float* scene_lines, int64_t n_lines);
int fdcm_featuremap_free(fdcm_featuremap* fm);
int fdcm_featuremap_get_info(const fdcm_featuremap* fm, fdcm_featuremap_info* info);
int fdcm_featuremap_keys(const fdcm_featuremap* fm, float* keys /* depth floats, ascending */);
/* Slice k as the reference stores it: RawImage<float>(H, W) column-major, (y,x) at x*H + y. */
int fdcm_featuremap_slice(const fdcm_featuremap* fm, int64_t k, float* out_host);
/* Whole volume on the device (read-only view, valid until free/rebuild).  Layout: 4 neighbouring x are
 * interleaved so that a 64-byte sector holds 4 x by 4 y pixels (the search's gathers step about one pixel per
 * translation in any direction), and slices are a little longer than their pixels (power-of-two slice
 * strides would put one pixel of every slice on the same memory channel): pixel (k, x, y) is element
 * k * floats_per_slice + ((x/4) * H + y) * 4 + x%4, floats_per_slice from fdcm_featuremap_device_volume_stride. */
int fdcm_featuremap_device_volume(const fdcm_featuremap* fm, const float** device_ptr);
int fdcm_featuremap_device_volume_stride(const fdcm_featuremap* fm, int64_t* floats_per_slice);
int fdcm_featuremap_last_timing(const fdcm_featuremap* fm, fdcm_build_timing* t);
/* The device-side times of fdcm_build_timing / fdcm_search_timing cost a HIP event between every two kernels of the build and
 * around the search (3 - 5 us each on a blocking frame).  on = 1 (default): per-stage times.  on = 2: events around the
 * build and around the search only (total_ms and the search's kernel_ms; the stage fields are 0).  on = 0: no events: the
 * timings carry the host time and the counters (candidates, evaluations) only, every device time is 0.
 * No counterpart in the reference (it has no timers). */
int fdcm_featuremap_stage_timing(fdcm_featuremap* fm, int on);
/* Dt3Cpu(dt3map, sceneTranslation, featureSize) constructor (dt3cpu.h:55-58): adopt caller slices. */
int fdcm_featuremap_from_slices(const float* keys, int64_t depth, const float* volume_host /* [k][x][y] */,
                                int64_t width, int64_t height, const float scene_translation[2],
                                fdcm_featuremap** out);
/* Test hook: stop the build after stage 1 (distance transform), 2 (propagation) or 3 (all). */
int fdcm_featuremap_build_staged(const float* scene_lines, int64_t n_lines, int64_t depth, float dt3_coeff,
                                 float padding, int distance, int stop_after, fdcm_featuremap** out);

/* ---- the feature-map plug-in seam: what a FeatureMapInstance specialises besides getFeatureSize
 *      (featuremap.h:27-52; FeatureMapModel<T> forwards to them, featuremap.h:80-92) and what every optimiser of the
 *      reference calls (defaultoptimize.cpp:26,49-64, batchoptimize.cpp:27,58-62).  With these three a reference
 *      build can wrap a handle in its own type-erased FeatureMap and run ANY of its optimisers on the HBM volume
 *      (INTEGRATION.md).  Both are batched: one kernel launch per call. ---- */
/* minmaxTranslation<Dt3Cpu>, dt3cpu.cpp:30-75,119-124: the admissible multiplier interval {negative, positive} of
 * align_vec for a template (4 x n_lines, in scene coordinates: the feature map adds its scene translation itself).
 * {inf, inf} for a zero align_vec, {NaN, NaN} when the template's bounding box starts outside the feature map. */
int fdcm_featuremap_minmax_translation(const fdcm_featuremap* fm, const float* tmpl_lines, int64_t n_lines,
                                       const float align_vec[2], float out_minmax[2]);
/* the same for n_templates templates (lines concatenated, line_offsets in lines) with one align_vec each;
 * out_minmax: 2 floats per template */
int fdcm_featuremap_minmax_translation_batch(const fdcm_featuremap* fm, const float* tmpl_lines, const int64_t* line_offsets,
                                             int64_t n_templates, const float* align_vecs, float* out_minmax);
/* evaluate<Dt3Cpu>, dt3cpu.cpp:126-179: scores[t][j] = sum_i |I_bin(i)(p1_i + T + tr_j) - I_bin(i)(p2_i + T + tr_j)| for
 * every template t and each of its translations tr_j (x, y pairs; translation_offsets in translations, n_templates + 1
 * entries), T = the scene translation, coordinates truncated like cast<int>(), terms added in Eigen's sum() order:
 * the reference's bits.  scores_out: one float per translation, in input order.  A translation that puts an end
 * point outside the feature map scores NaN (the reference reads out of bounds there, dt3cpu.cpp:166-167). */
int fdcm_featuremap_evaluate(const fdcm_featuremap* fm, const float* tmpl_lines, const int64_t* line_offsets,
                             int64_t n_templates, const float* translations, const int64_t* translation_offsets,
                             float* scores_out);

/* ---- templates: the `templates` argument of search(), kept resident in HBM ---- */
int fdcm_templates_create(const float* lines, const int64_t* offsets /* n_templates+1, in lines */,
                          int64_t n_templates, fdcm_templates** out);
int fdcm_templates_free(fdcm_templates* t);
int fdcm_templates_count(const fdcm_templates* t, int64_t* n_templates, int64_t* n_lines);
/* getTemplateLengths, math.h:319-324 */
int fdcm_templates_lengths(const fdcm_templates* t, float* lengths /* n_templates */);

/* ---- search<DefaultMatch> with DefaultSearch(max_tmpl_lines, max_scene_lines) and
 *      DefaultOptimize / BatchOptimize(batch_size): defaultmatch.cpp:32-89 ----
 * Returns the matches in the reference's positional order (template order x search-combination
 * order x 2 alignments, candidates without a value skipped).  tmpl_idx is offset by
 * tmpl_index_base (0 for a single GPU; the shard's first template for sharded runs).
 * The limits behave as the reference's min(): max_tmpl_lines above a template's line count means all of its
 * lines (defaultsearch.cpp:38, size_t), max_scene_lines above n_scene_lines means every scene line
 * (defaultsearch.h:42-46; the reference's int casts make values >= 2^31 undefined there, here they also mean
 * "every line"). */
int fdcm_search(const fdcm_featuremap* fm, const fdcm_templates* templates, const float* scene_lines,
                int64_t n_scene_lines, int64_t max_tmpl_lines, int64_t max_scene_lines, int optimizer,
                int64_t batch_size, int32_t tmpl_index_base, fdcm_match** out, int64_t* n_out);
/* Same, results left on the device: out_device must hold fdcm_search_capacity() records;
 * *n_out receives the count (host).  For RCCL gathers without a host round trip. */
int fdcm_search_capacity(const fdcm_templates* templates, int64_t n_scene_lines, int64_t max_tmpl_lines,
                         int64_t max_scene_lines, int64_t* capacity);
int fdcm_search_device(const fdcm_featuremap* fm, const fdcm_templates* templates, const float* scene_lines,
                       int64_t n_scene_lines, int64_t max_tmpl_lines, int64_t max_scene_lines, int optimizer,
                       int64_t batch_size, int32_t tmpl_index_base, fdcm_match* out_device, int64_t* n_out);
int fdcm_search_last_timing(const fdcm_featuremap* fm, fdcm_search_timing* t);
void fdcm_matches_free(fdcm_match* m); /* match arrays are pinned host buffers from a pool inside the library */
/* Concatenate the valid records of n_blocks fixed-capacity blocks that sit back to back in device memory into a
 * library-owned host array, in block order.  A block is capacity_records + 1 records; the first int64 of its last
 * record holds its record count -- what a gather of fdcm_search_device outputs assembles when every rank appends its
 * count that way (openfdcm_amd/dist.py).  Queued on `stream` (a hipStream_t of the current device, NULL = the default
 * stream) behind whatever filled the blocks there; returns when the array is complete (a kernel writes it into pinned
 * memory: no copy command).  Release with fdcm_matches_free. */
int fdcm_blocks_to_host(const void* blocks_device, int32_t n_blocks, int64_t capacity_records, void* stream,
                        fdcm_match** out, int64_t* n_out);

/* ---- template shards over several GPUs of one node, from ONE process (the reference's own
 *      parallel seam is the per-candidate task loop, batchoptimize.cpp:102-114) ----
 * The template list is cut into contiguous index ranges, one per device; every device rebuilds the DT3 volume itself
 * from the scene lines (a long-lived host thread per device runs rebuild -> search) and the match records -- in top-k mode the
 * k best of every shard (penalise + stable sort on each device) -- travel to the first device in ONE grouped RCCL
 * send/recv per frame with exact sizes; no count exchange is needed because all shards live in this process.
 * fdcm_sharded_search returns what fdcm_search returns for the whole list on one device (same records, same order);
 * fdcm_sharded_search_topk returns what fdcm_topk returns for it.  devices = NULL means devices 0..n_devices-1.
 * RCCL (librccl.so.1) is bound at run time, only when n_devices > 1 or FDCM_SHARDED_ALWAYS_COLLECTIVE is set (then a
 * single shard also sends its records to itself through RCCL: a test hook for one-GPU machines).  Every entry point
 * leaves the caller's current device (the library's and HIP's) as it found it. */
#define FDCM_SHARDED_ALWAYS_COLLECTIVE 1
/* Test hook for one-GPU machines: `devices` may name a device more than once.  Shards that live on the first shard's
 * device hand their records over with device copies (RCCL cannot put one device into a communicator twice), everything
 * else -- ranges, one worker per shard and frame slot, offsets into the gathered array, the top-k merge -- is the
 * multi-device code. */
#define FDCM_SHARDED_ALLOW_SAME_DEVICE 2
typedef struct fdcm_sharded fdcm_sharded;
int fdcm_sharded_create(const int* devices, int n_devices, const float* tmpl_lines, const int64_t* offsets /* n_templates+1 */,
                        int64_t n_templates, int64_t depth, float dt3_coeff, float padding, int distance, int flags,
                        fdcm_sharded** out);
int fdcm_sharded_search(fdcm_sharded* s, const float* scene_lines, int64_t n_scene_lines, int64_t max_tmpl_lines,
                        int64_t max_scene_lines, int optimizer, int64_t batch_size, fdcm_match** out, int64_t* n_out);
/* penalty: -1 (none), FDCM_DEFAULT_PENALTY or FDCM_EXPONENTIAL_PENALTY(tau); at most k records per shard cross the links */
int fdcm_sharded_search_topk(fdcm_sharded* s, const float* scene_lines, int64_t n_scene_lines, int64_t max_tmpl_lines,
                             int64_t max_scene_lines, int optimizer, int64_t batch_size, int penalty, float tau, int64_t k,
                             fdcm_match** out, int64_t* n_out);
/* Frames in flight (like fdcm_pipeline_* on one device): the engine keeps n_frames frame slots per device, each with its
 * own feature map and a long-lived host thread.  fdcm_sharded_submit copies the scene lines, hands the frame to the
 * workers of every device and returns a ticket (tickets count up from 0; ticket t uses slot t % n_frames, so at most
 * n_frames tickets may be outstanding); fdcm_sharded_wait blocks until the frame is complete on every device, runs its
 * exchange and returns exactly what the blocking call returns -- while the workers compute the frames submitted
 * after it.  fdcm_sharded_search / _search_topk are submit + wait.  One caller thread at a time per engine; n_frames
 * is 1 after create and may be changed (1..16) while no frame is in flight. */
int fdcm_sharded_set_frames_in_flight(fdcm_sharded* s, int n_frames);
/* What the devices share out.  FDCM_SHARD_TEMPLATES (the default): every frame runs on every device,
 * each over its contiguous template range, one exchange per frame -- the way to shorten ONE frame, bounded by the build that
 * every device repeats.  FDCM_SHARD_FRAMES: ticket t runs WHOLE on device t % n_devices over the whole template list (uploaded
 * to every device by this call), slot (t / n_devices) % n_frames there; no exchange at all, results exactly those of
 * fdcm_search / fdcm_topk on one device -- the way to raise the throughput of a STREAM of frames (n_devices * n_frames tickets may be
 * outstanding; wait for them in any order, in submission order to keep every device busy).  Only while no frame is in
 * flight; tickets restart at 0 after a change of mode. */
#define FDCM_SHARD_TEMPLATES 0
#define FDCM_SHARD_FRAMES 1
int fdcm_sharded_set_mode(fdcm_sharded* s, int mode);
int fdcm_sharded_submit(fdcm_sharded* s, const float* scene_lines, int64_t n_scene_lines, int64_t max_tmpl_lines,
                        int64_t max_scene_lines, int optimizer, int64_t batch_size, int64_t* ticket);
int fdcm_sharded_submit_topk(fdcm_sharded* s, const float* scene_lines, int64_t n_scene_lines, int64_t max_tmpl_lines,
                             int64_t max_scene_lines, int optimizer, int64_t batch_size, int penalty, float tau, int64_t k,
                             int64_t* ticket);
int fdcm_sharded_wait(fdcm_sharded* s, int64_t ticket, fdcm_match** out, int64_t* n_out);
/* devices / shard_begin (n_devices + 1 entries: shard i holds templates [shard_begin[i], shard_begin[i+1])) may be NULL;
 * collectives = grouped send/recv operations issued so far, bytes_moved = bytes they carried. */
int fdcm_sharded_info(const fdcm_sharded* s, int* n_devices, int* devices, int64_t* shard_begin, int64_t* collectives,
                      int64_t* bytes_moved);
int fdcm_sharded_last_timing(const fdcm_sharded* s, int shard, fdcm_build_timing* bt, fdcm_search_timing* st);
int fdcm_sharded_free(fdcm_sharded* s);

/* ---- frame pipeline (throughput extension; the reference has no counterpart: its callers loop over frames
 *      and each search() blocks, python/src/matching.cpp:283-300).  One frame at the reference's sizes is
 *      latency bound on this GPU, so n_slots frames are kept in flight: each slot owns a feature map (own
 *      HBM volume, workspaces, HIP stream) and a host worker thread that runs fdcm_featuremap_rebuild +
 *      fdcm_search for the frames it is handed.  Tickets count up from 0; ticket t runs on slot t % n_slots,
 *      so at most n_slots tickets may be outstanding.  Results per frame are exactly those of the blocking
 *      calls.  One caller thread at a time per pipeline. ---- */
typedef struct fdcm_pipeline fdcm_pipeline;
int fdcm_pipeline_create(int64_t depth, float dt3_coeff, float padding, int distance, const fdcm_templates* templates,
                         int64_t max_tmpl_lines, int64_t max_scene_lines, int optimizer, int64_t batch_size,
                         int32_t tmpl_index_base, int n_slots, fdcm_pipeline** out);
/* Copies the scene lines and returns at once.  out_device: NULL (matches are returned by wait as a host
 * array) or a device buffer of fdcm_search_capacity() records that must stay valid until the wait. */
int fdcm_pipeline_submit(fdcm_pipeline* p, const float* scene_lines, int64_t n_lines, fdcm_match* out_device,
                         int64_t* ticket);
/* Blocks until the frame is complete.  *out (host array, release with fdcm_matches_free) is set when the
 * frame was submitted without a device buffer; bt / st may be NULL. */
int fdcm_pipeline_wait(fdcm_pipeline* p, int64_t ticket, fdcm_match** out, int64_t* n_out, fdcm_build_timing* bt,
                       fdcm_search_timing* st);
int fdcm_pipeline_slots(const fdcm_pipeline* p, int* n_slots);
int fdcm_pipeline_free(fdcm_pipeline* p); /* waits for frames in flight */

/* ---- ConcentricRangeStrategy (searchstrategies/concentricrange.h:73-84, concentricrange.cpp:29-60): the
 *      scene lines whose centre lies in the annulus low - FLT_EPSILON < r < high around `center`.  The
 *      strategy is DefaultSearch over those lines, so a caller searches with the filtered line array
 *      (search<DefaultMatch> only uses the geometry of the scene lines, defaultmatch.cpp:57-61). ---- */
int fdcm_filter_in_range(const float* lines, int64_t n_lines, const float center[2], float low_boundary,
                         float high_boundary, int64_t* out_indices, int64_t* n_out);

/* ---- tail: penalize (penaltystrategy.h) + sort_matches (matching.cpp:302-307), host side ---- */
int fdcm_penalize(int penalty, float tau, fdcm_match* matches, int64_t n, const float* template_lengths,
                  int64_t n_templates);
int fdcm_sort_matches(fdcm_match* matches, int64_t n);
/* sortMatches(matches, maxNumCandidates) (matchstrategy.h:52-55): std::partial_sort -- the min(max_num_candidates, n) best in
 * ascending score in front, the rest behind them in the order the algorithm leaves. */
int fdcm_partial_sort_matches(fdcm_match* matches, int64_t n, int64_t max_num_candidates);

/* ---- device tail: penalize + sort_matches + "the k best" on matches resident in HBM (the reference's callers do
 *      penalize(), sort_matches() and slice: README.md:71-72, python/src/matching.cpp:291-307).  matches_device:
 *      a buffer filled by fdcm_search_device (n records), or NULL for the matches of the last fdcm_search on
 *      `fm`.  penalty: FDCM_DEFAULT_PENALTY, FDCM_EXPONENTIAL_PENALTY (tau) or -1 for none.  Returns min(k, n)
 *      records in ascending penalised score, ties in positional order (the reference's std::sort leaves ties
 *      unspecified); scores are the reference's bits (denominators from the host libm, IEEE division on the
 *      device).  A record whose tmpl_idx - tmpl_index_base is outside the template set gets a NaN score and
 *      sorts last (fdcm_penalize on the host reports an error for the same input, like the reference's
 *      templatelengths.at()).  Release with fdcm_matches_free.  In sharded runs each rank sends its k best instead
 *      of all. ---- */
int fdcm_topk(fdcm_featuremap* fm, const fdcm_templates* templates, const fdcm_match* matches_device, int64_t n,
              int32_t tmpl_index_base, int penalty, float tau, int64_t k, fdcm_match** out, int64_t* n_out);

/* ---- the reference's line files (.lines / .scene / .tmpl): read / write of core/serialization.h:99-132 (Python: openfdcm.read
 *      / openfdcm.write, python/src/core.cpp:41-42).  Host only.  fdcm_lines_read hands out n lines as 4 floats each
 *      (x1 y1 x2 y2 = the 4 x N column-major LineArray), to be released with fdcm_lines_free; a missing file, a file that is
 *      not a line file and an unknown line data format are FDCM_EINVAL with the reference's message in fdcm_last_error().
 *      fdcm_lines_write replaces an existing file, as the reference does. ---- */
int fdcm_lines_read(const char* path, float** lines, int64_t* n_lines);
int fdcm_lines_write(const char* path, const float* lines, int64_t n_lines);
void fdcm_lines_free(float* lines);

/* ---- host-side self checks (no GPU needed) ---- */
/* Compare the device-portable atanf restatement with this machine's libm atanf over the float
 * bit patterns first, first+stride, ... (count values); returns the number of mismatches. */
int64_t fdcm_selftest_atanf(uint32_t first, uint32_t stride, uint64_t count);
/* Where the searches of this process take the orientation bins of the aligned template lines from (closestOrientation,
 * dt3cpu.h:93-114, on atanf, math.h:295-299): 0 = the device's restatement of atanf (it agrees with this machine's libm
 * on the sample the first search checks), 1 = this machine's libm on host threads (the sample disagreed -- another
 * glibc -- or FDCM_FORCE_HOST_BINS=1; slower: every search recomputes align / transform / atanf of every candidate
 * line on the host, and says so once on stderr).  Decided at the first call of this function or of a search. */
int fdcm_orientation_bins_mode(void);
/* Column ranges the balanced L2 / L2^2 sweep cuts a row of a slice with n seeded columns into (1 .. 8; csrc/fdcm_sweep.h:
 * the kernel calls the same function).  FDCM_SWEEP_MINCOLS=1..64, the tests' switch, lowers the columns a range holds at
 * least from 16, so that small test images exercise all 8 ranges: this call lets a test see that the switch took. */
int fdcm_selftest_sweep_ranges(int n_seeded_columns);
/* Builds of this process whose L2 / L2^2 sweep took its workgroup launch order from the per-chunk times of the handle's
 * previous build (`from_history`) / from the host's proxy, i.e. a handle's first build of a shape (`from_proxy`); builds
 * small enough to be resident at once take no order and count in neither.  Lets a test see that a frame slot whose buffers
 * are reserved before every frame (fdcm_sharded_submit) keeps its history. */
int fdcm_selftest_sweep_order_counts(int64_t* from_history, int64_t* from_proxy);
/* Column ranges that waves of the balanced L2 / L2^2 sweep took over from slower ones, over all builds of this handle's
 * present scratch (dynamic cuts: a wave out of columns begins a new range in the middle of the longest stretch nobody has
 * started; by default only where one blocking build has the GPU to itself, FDCM_SWEEP_STEAL=<blocks> forces a threshold,
 * 0 = never).  Lets a test see that the path it means to exercise ran. */
int fdcm_selftest_sweep_steals(fdcm_featuremap* fm, int64_t* count);

#ifdef __cplusplus
}
#endif
#endif /* FDCM_H */
