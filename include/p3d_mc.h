/*
 * p3d_mc.h -- C ABI of the MI355X-native marching-cubes hot path (libp3dmc.so).
 *
 * This is the drop-in boundary beneath Primitive3D's pybind surface.  Every entry point names the
 * reference interface it replaces (paths are into the upstream repo lzhnb/Primitive3D):
 *
 *   prim3d::marching_cubes(const Tensor&, float, vector<float>, vector<float>)
 *       src/prim3d/Utility/marching_cubes.h:14-15, marching_cubes.cu:212-305
 *         :229-252  counters + count_vertices_faces_kernel + 2x .item()  -> p3d_mc_count + p3d_mc_read_counts
 *         :257-298  allocations + gen_vertices/gen_faces kernels + scale/offset epilogue -> p3d_mc_emit
 *   (the pybind module src/pybind/bindings.cpp:30 keeps its signature; it becomes a thin adapter that
 *   allocates torch tensors between the two phases -- see INTEGRATION.md.)
 *
 * Conventions
 *   - All pointers except `counts` outputs are DEVICE pointers owned by the caller.  The library keeps two small
 *     things of its own per device / stream: a ring of pre-cleared 4 KiB blocks for the streaming kernel's output
 *     cursors and a 4 KiB pinned host mailbox through which the kernels report totals.  No torch types cross
 *     this boundary.
 *   - Threads: every entry point may be called concurrently from several host threads, on different streams or on
 *     ONE stream (each call with its own workspace and output buffers; see p3d_mc_release_stream for what is kept per
 *     stream); the launches of a whole-grid
 *     p3d_mc_extract_fused are enqueued under a per-stream lock, so calls sharing a stream queue up whole.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All calls enqueue work on
 *     it and return immediately; p3d_mc_read_counts waits for the totals only (see there), not for the stream.
 *   - Grid layout is the reference's: contiguous [rx][ry][rz], z fastest (marching_cubes.cu:20).
 *   - Return value 0 on success, negative P3D_E* on failure; p3d_last_error() gives a thread-local
 *     message.
 *   - Vertex order and face order are unspecified (as in the reference, whose order is atomicAdd
 *     arrival order, marching_cubes.cu:104,199).  Here the face order is deterministic for a given build and
 *     input; the one-pass call's vertex order is arrival order inside 32 regions (the counting call's is
 *     deterministic).
 */
#ifndef P3D_MC_H_
#define P3D_MC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P3D_MC_ABI_VERSION 10

/* dtype of the scalar field */
#define P3D_F32 0
#define P3D_F16 1

/* error codes */
#define P3D_OK 0
#define P3D_EINVAL (-1)   /* bad argument (null pointer, dims < 1, unknown dtype) */
#define P3D_ERANGE (-2)   /* problem too large for int32 vertex ids / workspace math */
#define P3D_EHIP (-3)     /* a HIP runtime call failed */

/* Slab description for multi-GPU runs (axis-0 slabs, one halo plane).
 * A single-GPU call passes slab = NULL. */
typedef struct p3d_mc_slab {
    int32_t halo_last_plane; /* 1: plane rx-1 of `grid` is a halo copy of the next rank's first
                                plane: its in-plane (axis 1/2) edges are owned by that rank, so no
                                vertices are emitted for them; their index records are imported by writing
                                what the owner exported (p3d_mc_export_plane_records) to the location
                                p3d_mc_plane_records returns for plane rx-1. */
    int32_t part;            /* p3d_mc_extract_fused only.  0: the whole slab in one call.
                                1: stream planes [0, split_plane) and return (no finalize, no face pass) -- lets
                                   the interior run while the halo plane is still in flight;
                                2: continue with planes [split_plane, rx), then finalize and count faces;
                                3: stream planes [split_plane, rx) (all of them if split_plane is 0) and write V and the
                                   id prefixes into the workspace header, but do not finalize: collectives that
                                   only need those (all-gather of V = the first int64 of ws, export of the first
                                   plane's records) can be enqueued now and travel while part 4 runs;
                                4: face count and the first part of the vertex compaction; V and F to the host;
                                5: faces (with the halo plane's imported records and the id bases) and the rest of
                                   the vertex compaction in one launch -- what parts 0/2 do for a slab without a
                                   halo plane.  (After parts 0/2 with a halo plane p3d_mc_emit writes the faces.)
                                6: like 5, but with the WHOLE vertex compaction: for a caller that gave part 4 no vertex
                                   buffer (vertices = NULL, cap_vertices = 0) because it sizes its outputs from the totals
                                   part 4 reports -- stream once (part 3, split_plane 0), count (part 4), read V and F,
                                   allocate exactly, emit (part 6): the reference's count -> allocate -> emit order
                                   (marching_cubes.cu:242-287) with ONE pass over the field.  Part 6 may also follow a
                                   whole call (part 0 / slab = NULL) on the same workspace and scratch whose output
                                   buffers were too small while the scratch was not (p3d_mc_read_counts: totals above
                                   the capacities, bit 0 clear): faces and vertices are written again, into the larger
                                   buffers, without a second pass over the field.
                                All parts of one extraction must be given the same workspace, grid shape, stream and
                                scratch buffer (the vertex buffer from part 4 on, if part 4 was given one).  The library
                                CHECKS the order (per workspace; host side, before anything is launched) and returns
                                P3D_EINVAL with a message for every successor this table does not have:
                                     call                      legal after (on the same workspace)
                                     0, 1, 3 with split 0      anything (starts a new extraction; each takes the next of the
                                                               stream's 16 pre-cleared cursor blocks -- part 1 HOLDS its
                                                               block for parts 2 / 3: other starts step over it, however
                                                               many; with 15 blocks held a start keeps its cursors in its
                                                               workspace header instead)
                                     2                         1 (same split_plane)
                                     3 with split_plane > 0    1 (same split_plane)
                                     4                         3
                                     5                         4 (the vertex buffer part 4 was given; none if it had none)
                                     6                         4, or a finished extraction (0, 2, 5, 6)
                                     p3d_mc_emit               p3d_mc_count[_scan], 4, or a finished extraction
                                (p3d_mc_count, p3d_mc_count_scan and the batched entry also start anew; the state diagram is in
                                INTEGRATION.md.)  The reference's boundary has no such state: marching_cubes.h:14-15. */
    int64_t vertex_id_base;      /* added to every locally owned vertex id written into faces */
    int64_t halo_vertex_id_base; /* added to the imported records of the halo plane */
    int64_t x_origin;            /* global axis-0 index of local plane 0: vertex x = float(x_origin + x) + dt */
    int64_t split_plane;         /* see `part` */
    const int64_t* rank_counts;  /* optional DEVICE pointer to the all-gathered vertex counts of all ranks ([world]
                                    int64).  When set, the face kernel derives the two bases itself
                                    (vertex_id_base = sum of the counts of ranks < rank, halo base = that + this
                                    rank's count) and ignores the two fields above: the host never waits for the
                                    other ranks' counts, the all-gather result stays on the device. */
    int32_t rank;                /* index of this rank in rank_counts: 0..63 (one lane of a wave loads one rank's count) */
    int32_t rank_counts_stride;  /* int64 elements between two ranks' counts (0 = 1): the multi-GPU wrapper all-gathers
                                    the first three header words of every rank's workspace (V, -, flags), so that every
                                    rank also learns every rank's overflow flags; the counts are then 3 apart */
    void* export_first_plane_to; /* part 3 only, optional DEVICE buffer of bytes_per_plane bytes (p3d_mc_plane_records): the
                                    dense vertex-id records of local plane 0 are written there by the launch that writes
                                    the header -- what p3d_mc_export_plane_records does as a launch of its own */
    int32_t defer_totals;        /* parts 4 and 5 (ABI v10): 1 = part 4 does not report V and F to the host (no totals launch
                                    behind the face count); the part 5 that follows -- it must be given the same 1 -- reports
                                    them from the first block of its face launch.  For a caller that enqueues part 5 with a
                                    capacity guess before it has seen the totals (SlabExtractor's device path): one launch
                                    less on the stream per extraction */
    int32_t reserved;            /* 0 */
    const uint32_t* region_first_rows; /* part 0 without a halo plane only (ABI v10), optional HOST pointer to 41 ascending row
                                    numbers, [0] = 0, [40] <= cap_vertices, at most 2^28 rows between two of them: a PREDICTED LAYOUT of the streaming kernel's 32
                                    output regions inside `vertices`.  Region r owns rows [ [r], [r+1] ): give it the rows it
                                    NEEDED in the last call on the shape (p3d_mc_read_counts_ex reports the 32 totals), without
                                    slack; behind the regions, [ [32+g], [33+g] ) is spill area g (g < 8: what the regions
                                    4g .. 4g+3 cannot hold; a tenth of the expected vertex count over the eight is plenty for
                                    a slowly changing field).  The kernel stores a vertex at its region's first row + slot; a
                                    wave-plane that does not fit into what is left of its region takes rows of its spill area.
                                    Rows are final unless they lie at or beyond V: those few are moved into the free rows below
                                    V and the face kernel translates their ids.  No scratch buffer (vertex_scratch may be NULL),
                                    no second trip for the other rows; a field extracted twice in a row moves nothing.  A spill
                                    area overflowing sets bit 2 of the flags: the vertex buffer is incomplete, call p3d_mc_emit
                                    with exactly sized buffers (a second pass over the field).  Behind such a call p3d_mc_emit
                                    takes both buffers (never one alone); part 6 is not available.  The rows of `vertices` at
                                    and beyond V are the library's until the call's work is done. */
} p3d_mc_slab;

/* Bytes of device scratch p3d_mc_count / p3d_mc_emit need for an [rx,ry,rz] grid.
 * Replaces: `vertex_grids` + `counters` (marching_cubes.cu:229-230, 257-259); about 0.31 B/voxel
 * (1 bit/voxel of sign words, 8 B per 64-voxel unit of vertex-id records, 4 B per unit of counts) + O(1)
 * instead of 12 B/voxel. */
int p3d_mc_workspace_bytes(int64_t rx, int64_t ry, int64_t rz, size_t* bytes);

/* Phase 1 (replaces the count_vertices_faces_kernel launch, marching_cubes.cu:242-249): ONE pass over the field that
 * classifies it against `thresh` (inside = value > thresh, strict), leaves the per-voxel sign bitfield, the per-unit
 * vertex-id records and the totals of the 32 output regions in `ws`, and counts vertices and triangles -- the streaming
 * kernel of p3d_mc_extract_fused in count-only form (no vertex is made) + its face count.  Totals stay on the device in
 * `ws` until p3d_mc_read_counts.  `slab` may be NULL (whole grid); of a slab only halo_last_plane is read.
 * (ABI v10.  Until v9 this name ran a classification pass, three scans and left scan-numbered ids: that is
 * p3d_mc_count_scan now.) */
int p3d_mc_count(const void* grid, int dtype, int64_t rx, int64_t ry, int64_t rz, float thresh,
                 const p3d_mc_slab* slab, void* ws, void* stream);

/* The same counts with DETERMINISTIC dense vertex ids (a classification pass, per-unit counts, a prefix scan): what a
 * caller runs when p3d_mc_read_counts reports bit 1 (a region of the one-pass kernels numbered more than 2^26 vertices),
 * and what the parity tests use to get per-vertex edge keys out of p3d_mc_emit.  About 3x the time of p3d_mc_count.
 * p3d_mc_emit behind it writes the vertices by id (gather emitter). */
int p3d_mc_count_scan(const void* grid, int dtype, int64_t rx, int64_t ry, int64_t rz, float thresh,
                      const p3d_mc_slab* slab, void* ws, void* stream);

/* Read of the totals (replaces the two .item() calls, marching_cubes.cu:251-252) of the last p3d_mc_count /
 * p3d_mc_extract_fused on this workspace.  It does NOT synchronise the stream: the kernels store the totals into
 * a pinned, host-coherent mailbox slot as soon as they are known and this call polls the slot, so it returns
 * while the remaining kernels of the call may still be running (results are complete in stream order).  Falls
 * back to a 24-byte copy + hipStreamSynchronize when the mailbox is unavailable (P3D_NO_MAILBOX=1, no pinned
 * memory, slot recycled by 64 newer calls).  num_faces is triangles, not indices.  scratch_overflow (nullable)
 * receives two flags of the last p3d_mc_extract_fused:
 *   bit 0  the scratch buffer was too small for some output region: the vertex buffer is incomplete and
 *          p3d_mc_emit must be used to rewrite it (ids and counts stay valid);
 *   bit 1  one of the 32 regions received more than 2^26 vertices: the ids handed out by the one-pass call are
 *          ambiguous (they are region * 2^26 + slot).  The COUNTS are still right; the caller must renumber with
 *          p3d_mc_count_scan (dense ids by prefix scan) and then call p3d_mc_emit. */
int p3d_mc_read_counts(const void* ws, int64_t* num_vertices, int64_t* num_faces, int32_t* scratch_overflow,
                       void* stream);
/* The same, and the totals of the streaming kernel's 32 output regions (region_totals: 32 int64, nullable) -- what a caller
 * lays the next call's regions out from (p3d_mc_slab.region_first_rows).  They come with the totals (same mailbox slot) after a
 * whole-grid p3d_mc_extract_fused (part 0) or p3d_mc_count; after other calls they are unspecified.  scratch_overflow also
 * carries bit 2 (4): a region outgrew its rows of the predicted layout. */
int p3d_mc_read_counts_ex(const void* ws, int64_t* num_vertices, int64_t* num_faces, int32_t* scratch_overflow,
                          int64_t* region_totals, void* stream);

/* Phase 2 (replaces gen_vertices_kernel, gen_faces_kernel and the epilogue, marching_cubes.cu:266-298):
 * write vertices [V,3] f32 already mapped to the bounding box (v * scale + lower, scale as in
 * :293-297 including the upper[2]-lower[1] term of :295) and faces [F,3] i32.  Must follow finished counts on the same
 * grid / ws / stream (p3d_mc_count, p3d_mc_count_scan, part 4, or a whole p3d_mc_extract_fused).  cap_vertices / cap_faces
 * are the capacities of the two buffers
 * in elements (rows); nothing is written past them, and a capacity that is too small is not an error
 * here (the caller compares the counts it read with the capacities it gave).
 * How: for a whole grid (slab = NULL) whose vertices AND faces are asked for, behind counts of the one-pass kernels, a
 * SECOND streaming pass over the field (as the reference reads it a second time in gen_vertices_kernel): the 32 region
 * totals of the counting pass depend on the launch geometry alone, so every region is written at its final rows of the
 * caller's buffer -- no scratch, no copy; vertex ids are handed out anew and the faces use them.  In every other case ids
 * handed out earlier must survive (a slab whose first plane's records are with the neighbour already; vertices or faces
 * alone; after p3d_mc_count_scan; with vertex_keys) and a gather emitter writes the vertices by id.
 * vertex_keys (nullable) receives per-vertex edge keys voxel_linear*3+axis (int64, debug/parity output, local slab
 * indexing).
 * The same-stream rule of the order check applies: p3d_mc_emit must be given the stream of the call that made the counts
 * (the workspace is read in stream order; a host-side p3d_mc_read_counts in between does not replace that). */
int p3d_mc_emit(const void* grid, int dtype, int64_t rx, int64_t ry, int64_t rz, float thresh,
                const float lower[3], const float upper[3], const int64_t full_res[3],
                const p3d_mc_slab* slab, void* ws, float* vertices, int64_t cap_vertices,
                int32_t* faces, int64_t cap_faces, int64_t* vertex_keys, void* stream);

/* One-pass variant of count+emit for callers that can guess the output size (steady-state use: the
 * same grid shape every frame).  Reads the field ONCE: classification, vertex ids and vertex
 * emission happen in the same streaming kernel, faces follow from the sign bitfield.  Vertices are
 * first written into `vertex_scratch` ([scratch_rows,3] f32, caller-owned, not needed afterwards):
 * it is cut into 32 regions that fill independently (one output cursor per XCD group and wave, so
 * no single atomic address serialises the chip), then copied back to back into `vertices`; size it
 * about 1.25x the expected vertex count.  At most cap_vertices / cap_faces rows are written; the
 * true totals are always computed and are read with p3d_mc_read_counts afterwards.  If a total
 * exceeds its capacity, or a scratch region overflowed, the caller allocates exact buffers and calls
 * p3d_mc_emit on the same ws (no re-count needed: ids already assigned stay valid).
 * cap_vertices = cap_faces = 0 (scratch may be NULL) makes this a pure count.  With slab->halo_last_plane the face pass
 * is skipped (the caller imports the halo records first, then calls p3d_mc_emit with cap_vertices=0).
 * Replaces the same reference code as p3d_mc_count + p3d_mc_emit (marching_cubes.cu:229-298). */
int p3d_mc_extract_fused(const void* grid, int dtype, int64_t rx, int64_t ry, int64_t rz, float thresh,
                         const float lower[3], const float upper[3], const int64_t full_res[3],
                         const p3d_mc_slab* slab, void* ws, float* vertices, int64_t cap_vertices,
                         float* vertex_scratch, int64_t scratch_rows, int32_t* faces, int64_t cap_faces,
                         void* stream);

/* Batched one-pass extraction: `nitems` grids of ONE shape [rx,ry,rz], back to back in memory ([B,rx,ry,rz]
 * contiguous), extracted by one streaming launch, one counting launch and one face launch for the whole batch -- a
 * 256^3 grid alone cannot fill 256 CUs, so per-item calls are bound by launch ramp and tail (BASELINE.json config 5:
 * 32 x 256^3 fp16 density grids).  The reference has no counterpart (its entry rejects 4-D input,
 * marching_cubes.cu:219); every item gives exactly the mesh p3d_mc_extract_fused gives for it alone:
 *   vertices  [sum V_b, 3]  item b's rows are [item_offsets[b], item_offsets[b+1])
 *   faces     [sum F_b, 3]  item b's rows are [item_offsets[B+1+b], item_offsets[B+2+b]), vertex ids LOCAL to the item
 *   item_offsets            DEVICE memory, 2*(B+1) int64, written by the last launch
 * Capacities, scratch (32 regions PER ITEM: size it about 1.25x the expected total) and the overflow protocol are
 * those of p3d_mc_extract_fused: read the totals and flags with p3d_mc_read_counts; if something did not fit, call
 * again with exact capacities (or fall back to per-item calls).  Workspace: p3d_mc_workspace_bytes_batched. */
int p3d_mc_workspace_bytes_batched(int64_t nitems, int64_t rx, int64_t ry, int64_t rz, size_t* bytes);
int p3d_mc_extract_fused_batched(const void* grids, int dtype, int64_t nitems, int64_t rx, int64_t ry, int64_t rz,
                                 float thresh, const float lower[3], const float upper[3], void* ws, float* vertices,
                                 int64_t cap_vertices, float* vertex_scratch, int64_t scratch_rows, int32_t* faces,
                                 int64_t cap_faces, int64_t* item_offsets, void* stream);

/* Test hook: where the sign bitfield (u64 per 64-voxel unit, unit u = (x*ry+y)*ncz + c) and the
 * vertex-id records ({u32 base, u32 offY | offZ<<16} per unit) live inside `ws`, so a test can
 * rebuild the vertex-id -> edge-key map on the host and canonicalise a mesh (the reference has no
 * counterpart; its ids live in vertex_grids, marching_cubes.cu:257-259). */
int p3d_mc_debug_layout(int64_t rx, int64_t ry, int64_t rz, size_t* off_bits, size_t* off_records,
                        int64_t* num_units, int32_t* chunks_per_row);

/* Multi-GPU helpers.  A rank's halo plane (local plane rx-1, p3d_mc_slab.halo_last_plane) is the next rank's plane 0:
 * that rank exports the vertex-id records of its plane 0 in dense form (p3d_mc_export_plane_records: `out` receives
 * bytes_per_plane bytes; after a one-pass call the records inside ws are still region-relative, the export
 * translates them), they travel over RCCL send/recv, and the receiver stores them at the location
 * p3d_mc_plane_records returns for its plane rx-1 before it calls p3d_mc_emit for its faces.  plane is a local
 * axis-0 index. */
int p3d_mc_plane_records(void* ws, int64_t rx, int64_t ry, int64_t rz, int64_t plane,
                         void** records, size_t* bytes_per_plane);
int p3d_mc_export_plane_records(const void* ws, int64_t rx, int64_t ry, int64_t rz, int64_t plane, void* out,
                                void* stream);

/* Measurement hooks (no reference counterpart; the reference's only instrumentation is the wall-clock
 * Timer of prim3d/misc/utils.py:41-116).  mode 0 = off, 1 = hipEvents around the dominant kernel
 * only, 2 = around every stage.  Events are recorded on the stream passed to count/emit.
 * p3d_mc_profile_read synchronises those events and returns the per-stage durations (ms, -1 = not
 * recorded) of the most recent call; it returns the number of stages (11; pass n >= 11).  The dominant
 * kernel's events ride on its own dispatch packet (hipExtLaunchKernel), so timing it does not perturb the stream.
 * The profiling state is one per process and NOT synchronised: enable it only while a single thread on a single
 * stream calls the library (bench.py, tools/); the extraction calls themselves are re-entrant with it off. */
int p3d_mc_profile_enable(int mode);
int p3d_mc_profile_read(float* stage_ms, int n);
const char* p3d_mc_profile_stage_name(int stage);

/* Environment (no reference counterpart: the reference's path reads none).  The library reads SIX variables, once, at its
 * first call -- every one a supported knob; none changes results, only how the work is launched:
 *   P3D_FUSED_BLOCKS   (2048)  the streaming launch aims at about this many blocks (x-slabs of at most 16 planes each)
 *   P3D_FUSED_XT       (rule)  planes per block of the streaming launch, overriding the rule above
 *   P3D_COMPACT_BLOCKS (256)   blocks that copy the 32 vertex regions to their dense place (a multiple of 32) ...
 *   P3D_COMPACT_EARLY  (3)     ... of which this many of every 8 ride with the counting launch, the rest with the face launch
 *   P3D_FACES_SPARSE   (rule)  the face launch's empty-tile pre-check: unset = chosen per launch from the face capacity,
 *                              0 = never, 1 = always (an object's SDF in a large box whose first call is already sparse)
 *   P3D_NO_MAILBOX     (0)     1 = p3d_mc_read_counts copies and synchronises instead of polling the pinned mailbox
 * The pybind adapter (libPrim3D) reads one more, P3D_MC_MODE = hinted (default) | scratch | exact.
 * p3d_mc_reload_tuning re-reads the first five (tests; not to be called while another thread is inside the library).
 *
 * Everything else that used to be readable from the environment -- the developer sweeps' launch knobs and the TEST HOOKS
 * (P3D_TEST_ID_LIMIT, P3D_TEST_INDEX_LIMIT, P3D_TEST_FAIL_AFTER_LEASE, P3D_NO_CHUNK_PRE, P3D_PARTS_RING, P3D_FUSED_XT_TAIL,
 * ...) -- is compiled in only with -DP3D_DEV_HOOKS=1 (primitive3d_amd/_build.py builds that variant as dev/libp3dmc.so
 * for the tests that need one).  p3d_mc_dev_hooks() says which variant is loaded: 0 = the default library, in which a
 * stray P3D_TEST_* in a deployment's environment changes nothing (tests/test_gpu_dev_hooks.py). */
int p3d_mc_reload_tuning(void);
int p3d_mc_dev_hooks(void);

/* Library-owned state (no reference counterpart: the reference keeps none, marching_cubes.cu:229-230 allocates its four
 * counters per call).  Per (device, stream) the library keeps a ring of pre-cleared blocks (cursors of the streaming
 * kernel: sixteen 4 KiB blocks) and per device a 4 KiB pinned mailbox; both are created on first use.
 *   p3d_mc_release_stream: frees what is kept for `stream` on the CURRENT device, after waiting for the work queued on it.
 *       Call it before destroying a stream the library was used on; a later call on that stream simply creates the state
 *       again.  Unknown streams are fine (returns 0).
 *   p3d_mc_shutdown: frees everything (all devices and streams, the mailboxes, the profiling events).  No other thread may
 *       be inside the library.  The library may be used again afterwards. */
int p3d_mc_release_stream(void* stream);
int p3d_mc_shutdown(void);

/* Counters of what the library has launched since it was loaded (no reference counterpart; tests and the bench read them
 * to see which path a call took -- nothing in the data path depends on them):
 *   out[0] streaming launches          out[1] streaming passes that stored through a region layout (p3d_mc_slab.region_first_rows)
 *   out[2] streaming passes of p3d_mc_extract_fused[_batched] that wrote or counted a whole grid / stack (part 0 or 2, 3)
 *   out[3] calls of p3d_mc_count / p3d_mc_count_scan
 *   out[4] emissions that had no streaming pass of their own (p3d_mc_slab.part = 6)
 *   out[5] (device, stream) pairs the library currently keeps a cursor ring for (p3d_mc_release_stream / p3d_mc_shutdown)
 *   out[6] bytes of device memory those rings hold (it returns to 0 when every ring has been freed)
 * Writes min(n, 7) values, returns how many. */
int p3d_mc_debug_counters(int64_t* out, int n);

const char* p3d_last_error(void);
int p3d_mc_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* P3D_MC_H_ */
