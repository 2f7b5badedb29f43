/*
 * Host kd-tree builder that emits the HBM layout of rt_hip.h directly.
 *
 * Same algorithm and fp32 arithmetic as the reference's builder
 * (rt_funcs_render_pipe.c:591-1209: scene box padded by 1, longest-axis
 * split, 32 SAH candidate planes, lowest-index argmin, stable partition by
 * triangle extent, leaf at <= maxPrims triangles or at maxDepth), therefore
 * the same tree, but
 *   - per-triangle extents are computed once instead of 32 times per level,
 *   - candidate counts come from one histogram pass (two binary searches
 *     per triangle) instead of 32 passes,
 *   - sub-trees are built by a pool of worker threads instead of eight
 *     threads created and joined per node,
 *   - maxDepth / maxPrims are run-time parameters (reference: 10 and 7).
 */
#ifndef RT_KDTREE_H
#define RT_KDTREE_H

#include "rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct _rt_kdtree_flat
{
	rt_hip_kdnode   *nodes;   uint64_t nNodes;
	rt_hip_trirec   *recs;    uint64_t nRecs;
	rt_hip_trishade *shade;   uint64_t nTris;
	rt_box bbox;
	int depth;                /* deepest leaf, root = 0 */
	/* pruned leaf lists only (else NULL / 0): recs then holds nRecs + nRecsFull records - the pruned lists
	 * the nodes point to, followed by every leaf's complete list - and leafFull, one entry per node, says
	 * where a leaf's complete list lies (rt_hip.h) */
	rt_hip_leaf_full *leafFull;
	uint64_t nRecsFull;
} rt_kdtree_flat;

/* Builder options.  With sahStop = 0, allAxes = 0 and pruneLeaves = 0 the tree
 * is the reference's for the given limits.  The extras (no reference
 * counterpart) trade tree equality for traversal speed:
 *   sahStop > 0  make a leaf when the best split costs >= sahStop x the cost
 *                of not splitting (the reference always splits down to
 *                maxPrims / maxDepth, which multiplies triangle references)
 *   allAxes = 1  evaluate the 32 candidates on all three axes instead of the
 *                longest one only
 *   pruneLeaves > 0  nodes and split planes stay the reference's; a leaf keeps
 *                only the triangles that touch its cell grown by pruneLeaves x
 *                the scene size on every side (exact triangle / box test in
 *                double precision), in their original order.  The reference
 *                files a triangle under every leaf its bounding box overlaps;
 *                a triangle that does not come near a cell cannot be hit
 *                inside the cell's distance interval, so closest hits and
 *                shadow sums are unchanged while fewer triangles are tested.
 *                (If the reference's partition dropped a triangle - one that
 *                lies in a split plane goes to neither child - the result of
 *                the walk depends on leaves that list a triangle they do not
 *                contain; such a tree keeps the reference's lists.  The same
 *                holds for the walk of a ray with a direction component of
 *                exactly zero that starts in a split plane: the complete lists
 *                are therefore kept behind the pruned ones, rt_kdtree_flat::
 *                leafFull, and the kernels walk such rays over those.)
 *                The margin has to cover the rounding of hit and split
 *                distances: 1e-5 is about 170 units in the last place of a
 *                coordinate at the scene's scale. */
typedef struct _rt_kdtree_build_opts
{
	int   maxDepth, maxPrims, threads;
	float sahStop;
	int   allAxes;
	float pruneLeaves;
	uint64_t stageTicket;      /* GPU builder only: the geometry was (partly) copied ahead, see rt_kdtree_device_stage; 0 = not */
} rt_kdtree_build_opts;

int  rt_kdtree_build_flat_ex( const rt_vertex *verts, const rt_triangle *tris, uint64_t vertexCount,
	uint64_t trianglesCount, const rt_kdtree_build_opts *opts, rt_kdtree_flat *out );

/* Returns 0 on success (1 bad arguments, 2 limits out of range, 3 vertex index
 * out of range, 4 tree too large: beyond the 30-bit links of a node, or beyond
 * a quarter of the machine's memory / 32 GiB while it is being built - triangles
 * that cannot be separated double their references with every level).
 * threads <= 0 picks the number of online CPUs (capped at 16).  Triangle vertex
 * indices must be global and < vertexCount. */
int  rt_kdtree_build_flat( const rt_vertex *verts, const rt_triangle *tris, uint64_t vertexCount,
	uint64_t trianglesCount, int maxDepth, int maxPrims, int threads, rt_kdtree_flat *out );
void rt_kdtree_flat_release( rt_kdtree_flat *t );

/* Expands the HBM layout back into the reference's flat interchange format
 * (48-byte nodes + uint64 index list); arrays are malloc'ed. */
int  rt_kdtree_flat_expand( const rt_kdtree_flat *t, rt_cl_kdtree_node **nodes, uint64_t *nodesCount,
	rt_ulong **primsIdx, uint64_t *primsIdxCount );

/* Converts a tree given in the interchange format into the HBM layout. */
int  rt_kdtree_flatten( const rt_cl_kdtree_node *nodes, uint64_t nodesCount, const rt_ulong *primsIdx,
	uint64_t primsIdxCount, const rt_vertex *verts, const rt_triangle *tris, uint64_t vertexCount,
	uint64_t trianglesCount, const rt_box *bbox, rt_kdtree_flat *out );

/*
 * The same builder on the GPU (no reference counterpart; same tree, bit for
 * bit, as rt_kdtree_build_flat for the same limits).  verts / tris are HOST
 * arrays as the reference API hands them over; the tree is left in DEVICE
 * memory (hipMalloc) in the layout of rt_hip.h, recs with one spare record at
 * the end.  rt_hip_adopt_tree (rt_hip.h) installs it in a context without a
 * copy; rt_kdtree_device_release frees a tree that was not adopted.
 */
typedef struct _rt_kdtree_device
{
	void *nodes;   uint64_t nNodes;      /* rt_hip_kdnode   */
	void *recs;    uint64_t nRecs;       /* rt_hip_trirec, nRecs + 1 entries allocated */
	void *shade;   uint64_t nTris;       /* rt_hip_trishade */
	rt_box bbox;
	int depth;
	uint32_t maxMaterial;                /* highest material slot any triangle references */
	/* pruned leaf lists only (else NULL / 0), as in rt_kdtree_flat: recs then holds nRecs + nRecsFull + 1 records */
	void *leafFull;                      /* rt_hip_leaf_full, one per node */
	uint64_t nRecsFull;
} rt_kdtree_device;

/* Returns 0 on success (1 bad arguments, 2 limits, 3 vertex index out of
 * range, 4 tree too large for 30-bit links, 10 HIP failure, see
 * rt_kdtree_device_last_error).  hipStream: stream to build on, NULL = default. */
int  rt_kdtree_build_device( const rt_vertex *verts, const rt_triangle *tris, uint64_t vertexCount,
	uint64_t trianglesCount, int maxDepth, int maxPrims, void *hipStream, rt_kdtree_device *out );
/* The same with the builder options of the host builder.  Of the extras only
 * pruneLeaves is available here (sahStop / allAxes: code 2): the leaf lists are
 * pruned in place after the build - a leaf's record range keeps its start, its
 * count drops, nRecs stays what it was - by the host builder's test applied to
 * the vertices the records hold, and not at all when the build dropped a
 * triangle that lies in a split plane; a copy of the complete lists is kept
 * behind them (leafFull, nRecsFull = nRecs).  No reference counterpart. */
int  rt_kdtree_build_device_ex( const rt_vertex *verts, const rt_triangle *tris, uint64_t vertexCount,
	uint64_t trianglesCount, const rt_kdtree_build_opts *opts, void *hipStream, rt_kdtree_device *out );
void rt_kdtree_device_release( rt_kdtree_device *t );
/* Geometry on its way to the device while the caller is still putting it together (no reference counterpart): copies
 * those of the leading nVerts vertices and nTris triangles of the arrays that earlier calls with this ticket have
 * not copied yet, asynchronously on hipStream, into the builder's input buffers of this thread and the current
 * device.  *ticket == 0 starts over and receives a fresh ticket; the build that is then given arrays with the same
 * leading entries and that ticket (rt_kdtree_build_opts.stageTicket) copies the rest only.  The entries handed
 * over must not change before that build returns.  When the buffers have gone to other geometry in the meantime
 * (another staging or build of the thread), *ticket comes back 0 and nothing was copied: the build copies everything,
 * as it does for a ticket it does not know.  Returns 0, or 1 / 10 as the builds do. */
int  rt_kdtree_device_stage( uint64_t *ticket, const rt_vertex *verts, uint64_t nVerts, const rt_triangle *tris,
	uint64_t nTris, void *hipStream );
/* copy a device tree to host arrays (malloc'ed; rt_kdtree_flat_release frees them) */
int  rt_kdtree_device_download( const rt_kdtree_device *t, rt_kdtree_flat *out );
const char *rt_kdtree_device_last_error( void );
/* the builder keeps its work buffers (per thread and device) for the next build; this frees them */
void rt_kdtree_device_trim( void );
/* The four device buffers of a tree that nothing reads any more (any of them may be NULL), for the next build of this
 * thread on the current device to write its result into, instead of hipFree: a frame loop that rebuilds its tree
 * spent a millisecond and more per frame freeing 0.4 GB.  One set is kept; rt_kdtree_device_trim frees it. */
void rt_kdtree_device_recycle( void *nodes, void *recs, void *shade, void *leafFull );

#ifdef __cplusplus
}
#endif
#endif
