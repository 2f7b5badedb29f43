/*
 * C-ABI HIP layer: the MI355X replacement for the reference's OpenCL
 * plumbing.  Plain pointers and sizes only; every entry point returns 0 on
 * success and a non-zero code on failure (text via rt_hip_last_error()).
 * The rt_render_pipe_* API (rt_funcs_render_pipe.h) is built on top of it;
 * callers that keep a scene resident (benchmarks, multi-GPU drivers) may use
 * it directly.
 *
 * What each group replaces in the reference (rt_funcs_render_pipe.c):
 *   rt_hip_create / _destroy          rt_init_opencl            :398-495
 *   rt_hip_upload_* / _set_camera     rt_init_buffers, rt_add_block and the
 *                                     clEnqueueWrite/Map calls  :127-396
 *   rt_hip_upload_tree                the memn/memi buffers     :1171-1207
 *   rt_hip_set_image / _set_tile      rt_render_pipe_calc_wg_size + the
 *                                     NDRange set-up            :33-94,:534-536
 *   rt_hip_render / _read_frame       rt_opencl_render          :497-573
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include "rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_hip_ctx rt_hip_ctx;

/* ---- HBM layouts -------------------------------------------------------- */

/*
 * 8-byte kd node, pre-order, left child = own index + 1.
 *   inner: a = bits of the fp32 split position, b = (rightIndex << 2) | axis
 *   leaf : a = index of the first triangle record, b = (count << 2) | 3
 */
typedef struct _rt_hip_kdnode { uint32_t a, b; } rt_hip_kdnode;

/* 48-byte leaf triangle record: one per (leaf, triangle) reference, stored in
 * leaf order, so a leaf scan is one contiguous read with no indirection.
 * e1 = v1 - v0 and e2 = v2 - v0 are the fp32 differences the intersection
 * test would otherwise compute itself; nv0 = -v0 (exact), so that src - v0 is
 * the single addition src + nv0.  The fields are interleaved in the pairs the
 * kernel's two-wide fp32 instructions take as operands: a 16-byte load puts
 * each pair into one aligned register pair. */
typedef struct _rt_hip_trirec
{
	float    nv0x, e1x, nv0y, e1y;
	float    nv0z, e1z, e2x, e2y;
	float    e2z;
	uint32_t tri;     /* index into the triangle / shading arrays */
	uint32_t mat;     /* material slot */
	uint32_t _pad;
} rt_hip_trirec;

/* 48-byte per-triangle shading record (vertex normals + material slot) */
typedef struct _rt_hip_trishade
{
	float    n0[3], n1[3], n2[3];
	uint32_t mat;
	uint32_t _pad[2];
} rt_hip_trishade;

/* Where a leaf's COMPLETE list lies when the list its node points to has been pruned (rt_kdtree.h, pruneLeaves):
 * one entry per node (inner nodes: zeros), first = index into the same record array, behind the pruned lists.
 * Only rays with a direction component of exactly zero read it: the reference gives such a ray NaN split
 * distances when its origin lies in a split plane, and what it then returns depends on leaves that list a
 * triangle they do not contain - so these rays are walked over the reference's lists. */
typedef struct _rt_hip_leaf_full { uint32_t first, count; } rt_hip_leaf_full;

typedef struct _rt_hip_stats
{
	uint64_t raysClosest;   /* closest-hit queries                         */
	uint64_t raysShadow;    /* shadow queries                              */
	uint64_t hits;          /* shaded hits                                 */
	uint64_t nodeVisits;    /* kd nodes touched   (RT_HIP_STATS only)      */
	uint64_t triTests;      /* triangle tests     (RT_HIP_STATS only)      */
	uint64_t spillPushes;   /* stack pushes that went past LDS (STATS)     */
	uint64_t maxStack;      /* deepest traversal stack seen (STATS)        */
	uint64_t nodeVisitsShadow;  /* part of nodeVisits spent in shadow queries (STATS) */
	uint64_t triTestsShadow;    /* part of triTests spent in shadow queries   (STATS) */
	/* lane-slots: 64 per wave-level iteration of the descend / leaf loop, so
	 * e.g. (nodeVisits - nodeVisitsShadow) / nodeSlots is the fraction of
	 * lanes doing useful work in the closest-hit descend loop (STATS) */
	uint64_t nodeSlots, triSlots, nodeSlotsShadow, triSlotsShadow;
	/* closest-hit queries and hits of each ray slot (slot 0 = primary rays) */
	uint64_t slotRays[RT_STACK_SIZE_MAX], slotHits[RT_STACK_SIZE_MAX];
} rt_hip_stats;

/* rt_hip_render flags */
#define RT_HIP_STATS    1u   /* run the counting variant of the kernels    */
#define RT_HIP_HITIDS   2u   /* record the primary ray's hit id per pixel  */
#define RT_HIP_TIMING   4u   /* bracket every launch with HIP events       */

/* hit ids: triangle index, -1 = miss, -2-i = sphere i */
#define RT_HIP_HIT_MISS (-1)

/* rt_hip_set_param keys */
#define RT_PARAM_STACK_SIZE        1   /* rays per pixel, 1..RT_STACK_SIZE_MAX, default 6 */
#define RT_PARAM_MAX_DEPTH         2   /* kd depth cap, default 10 (render pipe only)     */
#define RT_PARAM_MAX_PRIMS_IN_NODE 3   /* kd leaf size, default 7 (render pipe only)      */
#define RT_PARAM_BUILD_THREADS     4   /* host builder threads, 0 = automatic             */
#define RT_PARAM_TREE_CACHE        5   /* 1 = reuse the tree while geometry is unchanged  */
#define RT_PARAM_GPU_BUILD         7   /* render pipe only: kd-tree builder: 0 = host, 1 = GPU, 2 = GPU for scenes of 8192 triangles and more (default); same tree either way */
#define RT_PARAM_OVERLAP           6   /* 0 = run every pass on one stream, in order (default 1:
                                          shadow + lighting passes on auxiliary streams)  */

#define RT_PARAM_COOP_LIVE          8   /* a wave whose rays are down to this many (0..32, default 32) lets them share
                                          its lanes (cooperative walk: several leaves of a ray per round); 0 = every
                                          ray keeps one lane to the end.  Same image either way.                    */
#define RT_PARAM_COOP_STACK         9   /* traversal-stack entries per ray the cooperative walk may use (0..12, default
                                          12); a ray that needs more restarts its walk from the root.  Lower values
                                          exist to exercise that path.  Same image either way.                      */

#define RT_PARAM_LAUNCH_WAVES       10   /* waves per compute unit that one traversal pass occupies (1..20, default 16 =
                                          all that fit: a frame rendered on its own is fastest that way).  A caller
                                          that keeps several frames in flight (contexts sharing a scene) sets 8: a
                                          pass that takes every wave slot keeps the other frames' passes out until
                                          its waves retire.  Same image either way.                                */

#define RT_PARAM_PRUNE_LEAVES       11   /* 1 = trees built from now on list in a leaf only the triangles that touch its
                                          cell (margin RT_PRUNE_LEAVES_MARGIN x scene size; rt_kdtree.h, pruneLeaves):
                                          rt_hip_build_tree (default 0), and both builders of the render pipe
                                          (default 1).  0: the reference's leaf lists only.  Same frame, fewer triangle tests; the
                                          complete lists are kept behind the pruned ones for the rays whose walk
                                          depends on them (a direction component of exactly zero).                 */
#define RT_PRUNE_LEAVES_MARGIN 1e-5f

#define RT_HIP_MAX_TREE_DEPTH 64       /* deepest tree the traversal stack admits */

int  rt_hip_device_count( void );
const char *rt_hip_last_error( void );

rt_hip_ctx *rt_hip_create( int device );              /* NULL on failure */
void rt_hip_destroy( rt_hip_ctx *c );
/* run on a caller-owned hipStream_t (e.g. the framework's current stream);
 * NULL selects the context's own stream */
int  rt_hip_set_stream( rt_hip_ctx *c, void *hipStream );

int  rt_hip_upload_tree( rt_hip_ctx *c, const rt_hip_kdnode *nodes, uint64_t nNodes,
	const rt_hip_trirec *recs, uint64_t nRecs, const rt_hip_trishade *shade,
	uint64_t nTris, const rt_box *bbox, int depth );
/* The same for a tree with pruned leaf lists: recs holds nRecs + nRecsFull records (the pruned lists the nodes
 * point to, then the complete ones), leafFull one entry per node.  leafFull = NULL, nRecsFull = 0: as above. */
int  rt_hip_upload_tree_lists( rt_hip_ctx *c, const rt_hip_kdnode *nodes, uint64_t nNodes,
	const rt_hip_trirec *recs, uint64_t nRecs, uint64_t nRecsFull, const rt_hip_leaf_full *leafFull,
	const rt_hip_trishade *shade, uint64_t nTris, const rt_box *bbox, int depth );
/* Install a tree built on the device by rt_kdtree_build_device (rt_kdtree.h)
 * without copying: the context takes ownership of the three buffers and
 * clears *t.  Replaces rt_hip_upload_tree for geometry that changes every frame. */
struct _rt_kdtree_device;
int  rt_hip_adopt_tree( rt_hip_ctx *c, struct _rt_kdtree_device *t );
/* build on the context's device and stream from HOST geometry and install the result */
int  rt_hip_build_tree( rt_hip_ctx *c, const rt_vertex *verts, const rt_triangle *tris, uint64_t vertexCount,
	uint64_t trianglesCount, int maxDepth, int maxPrims );
/* Geometry on its way to the device while the caller is still adding to it (no reference counterpart; see
 * rt_kdtree_device_stage in rt_kdtree.h): what of the leading nVerts / nTris entries earlier calls have not copied
 * is copied now, asynchronously on the context's stream, and the next rt_hip_build_tree of this context - which must
 * be given arrays with the same leading entries, unchanged - copies only the rest.  Returns 0, 1 on failure, and 2
 * when what had gone ahead was lost (the builder's buffers of this thread went to other geometry in between, another
 * context's for instance): nothing is staged then, the build copies everything, and a further call would start over
 * from the first entry.  rt_hip_unstage_geometry: the next rt_hip_build_tree is for other geometry after all. */
int  rt_hip_stage_geometry( rt_hip_ctx *c, const rt_vertex *verts, uint64_t nVerts, const rt_triangle *tris, uint64_t nTris );
void rt_hip_unstage_geometry( rt_hip_ctx *c );
int  rt_hip_upload_materials( rt_hip_ctx *c, const rt_material *m, uint64_t n );
int  rt_hip_upload_spheres( rt_hip_ctx *c, const rt_sphere *s, uint64_t n );
int  rt_hip_upload_lights( rt_hip_ctx *c, const rt_point_light *l, uint64_t n );
int  rt_hip_set_camera( rt_hip_ctx *c, const rt_camera *cam );
/* Frames in flight (no reference counterpart: the reference blocks on every frame, rt_funcs_render_pipe.c:567-569):
 * a second context on the same GPU, on its own stream, renders frame i + 1 while frame i of the first drains its
 * sparse last passes.  It reads the tree, materials, spheres and lights of `from` (no copies: one working set in
 * the caches); camera, image size, tile and parameters stay per context.  `from` must outlive every frame of `c`,
 * and uploads into `from` must not run beside frames of `c`.  After ANY upload into `from` (tree, materials, spheres,
 * lights, an adopted tree) and after its destruction the copied device pointers may dangle: rt_hip_render on `c` then
 * returns an error until rt_hip_share_scene is called again. */
int  rt_hip_share_scene( rt_hip_ctx *c, const rt_hip_ctx *from );

/* full image size; allocates the per-pixel work buffers */
int  rt_hip_set_image( rt_hip_ctx *c, int w, int h );
/*
 * Multi-GPU split (no reference counterpart): the image is cut into bands of
 * bandH rows; this context renders bands part, part+nParts, ... and stores
 * them compactly (local row = (band/nParts)*bandH + row%bandH).
 * nParts = 1 renders the whole frame.
 */
int  rt_hip_set_tile( rt_hip_ctx *c, int bandH, int nParts, int part );
int  rt_hip_local_rows( rt_hip_ctx *c );              /* rows this context renders */
int  rt_hip_set_param( rt_hip_ctx *c, int key, int value );

/*
 * Enqueue one frame on the stream: ray generation, closest-hit / shading
 * passes for every ray slot, colour fold, 8-bit store.  d_out is a DEVICE
 * buffer of w * rt_hip_local_rows() rt_argb, or NULL to use the context's
 * own frame buffer; page-locked host memory (rt_hip_alloc_host, hipHostMalloc)
 * does as well - the last kernel then stores the pixels across PCIe, which is
 * how rt_render_pipe_draw fills screenData (valid after rt_hip_sync).
 * Asynchronous.
 */
int  rt_hip_render( rt_hip_ctx *c, void *d_out, unsigned flags );
/*
 * The same frame without a single allocation.  No counterpart in the
 * reference: it is for callers that cannot allocate - code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  The per-frame
 * work buffers come from one fixed array in device memory (RT_HIP_NOALLOC_BYTES,
 * held by every GPU the library runs on until the process ends) instead of
 * buffers sized for the image, hence the limits: at most
 * RT_HIP_NOALLOC_MAX_PIXELS pixels in this context's share of the image,
 * RT_HIP_NOALLOC_MAX_SLOTS rays per pixel and RT_HIP_NOALLOC_MAX_LIGHTS lights;
 * anything beyond is refused with an error.  d_out (device, w *
 * rt_hip_local_rows() rt_argb) is required, flags must be 0, tree / materials /
 * spheres / lights must have been uploaded (empty tables included), passes run
 * in order on the context's stream.  Every call on a GPU uses that GPU's one
 * array, so the entry point is SINGLE-FLIGHT PER GPU: frames enqueued on one
 * stream follow each other; a call on another stream (another context, or after
 * rt_hip_set_stream) while the previous frame has not completed is refused
 * with an error - synchronise first.  (Not checked under stream capture:
 * whoever replays captured frames keeps them apart.)  rt_hip_sync reports a
 * frame whose lists ran out of room, as for rt_hip_render.  Same pixels as
 * rt_hip_render.
 */
#define RT_HIP_NOALLOC_MAX_PIXELS 65536
#define RT_HIP_NOALLOC_MAX_SLOTS  6
#define RT_HIP_NOALLOC_MAX_LIGHTS 4
#define RT_HIP_NOALLOC_BYTES      ( 96u << 20 )
int  rt_hip_render_noalloc( rt_hip_ctx *c, void *d_out, unsigned flags );
int  rt_hip_sync( rt_hip_ctx *c );
int  rt_hip_read_frame( rt_hip_ctx *c, rt_argb *host );        /* internal frame buffer -> host, synchronous */
int  rt_hip_read_hit_ids( rt_hip_ctx *c, int32_t *host );      /* after RT_HIP_HITIDS */
/* Page-locked host memory for the frames that are read back (the reference's
 * blocking clEnqueueReadBuffer went into malloc'ed screenData,
 * rt_funcs_render_pipe.c:567-569): read-backs into it run at link speed
 * instead of through a staging copy.  NULL when the driver refuses. */
void *rt_hip_alloc_host( rt_hip_ctx *c, size_t bytes );
void rt_hip_free_host( rt_hip_ctx *c, void *host );
int  rt_hip_get_stats( rt_hip_ctx *c, rt_hip_stats *out );     /* synchronous */
/* Per-frame kernel time in ms, averaged over the frames rendered with
 * RT_HIP_TIMING since the previous call (at most the last 32): ms[0] =
 * closest-hit passes, ms[1] = shadow passes, ms[2] = lighting passes, ms[3] = fold + store, ms[4] = whole frame; launches[i] = launches
 * per frame.  Measured with HIP events on the stream each kernel runs on: one behind every kernel; a kernel's
 * start is the end of the last of its predecessors (the kernel before it on its stream, the kernel its stream
 * waited for), so a timed frame carries few markers more than an untimed one.
 * With RT_PARAM_OVERLAP = 1 passes share the chip, so ms[0..3] add up to more
 * than ms[4]; set it to 0 to time every pass running alone. */
int  rt_hip_kernel_times( rt_hip_ctx *c, float ms[5], int launches[5] );
/* The same plus wall[0..3]: per kernel (closest, shadow, lighting, fold) the wall-clock time per frame during which at
 * least one of its launches was running (length of the union of its launch intervals).  Unlike ms[], which adds up
 * launches that run side by side, wall[] never exceeds the frame time. */
int  rt_hip_kernel_times_ex( rt_hip_ctx *c, float ms[5], int launches[5], float wall[4] );
/* Every launch of the frames rendered with RT_HIP_TIMING since the last rt_hip_kernel_times[_ex] call, as
 * an interval startEnd[2 i], startEnd[2 i + 1] in ms after the start of the oldest such frame of `origin` (a
 * context on the same GPU, NULL = c): contexts that keep frames in flight side by side come onto one clock, and
 * the union of a kernel's intervals over all of them is the wall-clock time that kernel had work on the chip.
 * kind[i]: 0 closest-hit, 1 shadow, 2 lighting, 3 fold, 4 the whole frame.  Needs room (cap) for
 * frames * (3 * slots + 2) intervals; *count = intervals written.  Leaves the timings in place. */
int  rt_hip_kernel_intervals( rt_hip_ctx *c, rt_hip_ctx *origin, float *startEnd, int *kind, int cap, int *count );
/* Device self-test (no reference counterpart): the triangle test replaces the
 * IEEE division 1 / det by a hardware reciprocal plus one Newton step where
 * that is exact.  Evaluates both for all 2^32 float bit patterns; *mismatches
 * = inputs for which the value the kernels would use differs from the
 * division (must be 0), *fastPath = inputs that take the short path. */
int  rt_hip_selftest_rcp( rt_hip_ctx *c, uint64_t *mismatches, uint64_t *fastPath );

/* ---- multi-GPU frame assembly (no reference counterpart: the reference renders on one device) ------------------
 *
 * One process per GPU, the scene replicated, the image cut into bands of bandH rows dealt round-robin to the ranks
 * (rt_hip_set_tile).  rt_hip_gather_frame moves every rank's compact share to rank `dst` over RCCL (send / receive in
 * one group: every source uses its own xGMI link) and de-interleaves the shares into the frame there.  d_local must
 * hold rt_hip_gather_rows(h, bandH, nRanks) rows (the largest share, whole bands; the rows beyond this rank's own share
 * are never shown), d_frame (destination rank only) h rows, both of w rt_argb, in device memory.  Asynchronous on
 * `hipStream`; collective: every rank of the communicator must call it.  Errors: rt_hip_comm_last_error().
 * RCCL is opened with dlopen at the first rt_hip_comm_* call.                                                        */
typedef struct rt_hip_comm rt_hip_comm;
#define RT_HIP_COMM_ID_BYTES 128
const char *rt_hip_comm_last_error( void );
/* rank 0 makes an id and hands its bytes to the other ranks by whatever means the application has (file, socket, MPI) */
int  rt_hip_comm_unique_id( void *id128 );
rt_hip_comm *rt_hip_comm_create( int device, int nRanks, int rank, const void *id128 );     /* NULL on failure; collective */
void rt_hip_comm_destroy( rt_hip_comm *comm );
int  rt_hip_gather_rows( int h, int bandH, int nParts );
int  rt_hip_gather_frame( rt_hip_comm *comm, const void *d_local, void *d_frame, int w, int h, int bandH, int dst, void *hipStream );
/* the second half alone, for callers with their own transport: d_stage holds nParts shares of
 * rt_hip_gather_rows() rows each, share r first band = band r of the image */
int  rt_hip_deinterleave( const void *d_stage, void *d_frame, int w, int h, int bandH, int nParts, void *hipStream );

#ifdef __cplusplus
}
#endif
#endif
