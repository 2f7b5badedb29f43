/*
 * CPU oracle: TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's hot path (host SAH kd-tree build
 * + the per-pixel `raytrace` OpenCL kernel).  Only tests/, the smoke check
 * and the benchmark's cpu_baseline leg may load this library; the product
 * library (librt_amd.so) never does.
 *
 * Parity status
 *   - kd-tree builder: PINNED.  Checked node-for-node against the
 *     reference's own rt_kdtree_make_childs / rt_kdtree_pack_to_buffer
 *     compiled from /root/reference (oracle/_ref/librt_ref.so) and against
 *     the committed dumps in tests/golden/.
 *   - device kernel (ray generation, traversal, intersection, shading):
 *     PARITY UNPINNED.  The reference ships no golden vectors and its
 *     OpenCL kernel cannot be executed in the build container (no OpenCL
 *     device, no CPU OpenCL runtime).  The restatement is cross-checked by
 *     an independent tree-free brute-force closest hit and by hand-derived
 *     known-answer cases only.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include "rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct _orc_stats
{
	uint64_t raysClosest;   /* invocations of the closest-hit query          */
	uint64_t raysShadow;    /* invocations of the shadow (alpha) query       */
	uint64_t nodeVisits;    /* kd nodes touched (inner + leaf), both queries */
	uint64_t triTests;      /* ray/triangle tests, both queries              */
	uint64_t hits;          /* shaded hits                                   */
	uint64_t staleHits;     /* closest-hit queries that ended in the
	                           "hit flagged but no valid t" state            */
	uint64_t nodeVisitsShadow;   /* share of nodeVisits in shadow queries    */
	uint64_t triTestsShadow;     /* share of triTests in shadow queries      */
} orc_stats;

/* hit-id map encoding (primary ray only) */
#define ORC_HIT_MISS        (-1)
#define ORC_HIT_SPHERE(i)   (-2 - (int32_t)(i))

/*
 * Restatement of rt_kdtree_build + rt_kdtree_make_childs +
 * rt_kdtree_compute_sah + rt_kdtree_pack_to_buffer
 * (reference rt_funcs_render_pipe.c:591-1209).
 * maxDepth / maxPrims are MAX_DEPTH / MAX_PRIMS_IN_NODE made run-time.
 * Outputs are malloc'ed; release with orc_free.  Returns 0 on success.
 */
int orc_kdtree_build( const rt_vertex *verts, const rt_triangle *tris,
	uint64_t vertexCount, uint64_t trianglesCount, int maxDepth, int maxPrims,
	rt_box *bbox, rt_cl_kdtree_node **nodes, uint64_t *nodesCount,
	uint64_t **primsIdx, uint64_t *primsIdxCount );

void orc_free( void *p );

/*
 * Restatement of the `raytrace` kernel (reference CL/rt_raytrace.cl:6-89)
 * run for every pixel (x,y) with y = y0, y0+yStep, ... and x = x0, x0+xStep...
 * The first twelve parameters are the kernel's twelve arguments.
 * stackSize = STACK_SIZE (rays per pixel), kdStackCap = KDTREE_DEPTH.
 * hitId (nullable, w*h) receives the primary ray's hit id.
 * Pixels that are not sampled are left untouched in `o` and `hitId`.
 */
void orc_raytrace( const rt_raytrace_args *a, const void *prims,
	const rt_prim_desc *pd, const void *lights, const rt_light_desc *ld,
	const rt_camera *cams, const rt_material *materials,
	const rt_triangle *triangles, const rt_vertex *verticies,
	const rt_cl_kdtree_node *kdtreeNodes, const uint64_t *kdtreePrimsIdx,
	rt_argb *o,
	int stackSize, int kdStackCap,
	int x0, int xStep, int y0, int yStep,
	int32_t *hitId, orc_stats *stats, int nThreads );

/*
 * Tree-free closest hit of the primary rays over ALL triangles and spheres
 * (the reference keeps this loop as a commented-out cross-check,
 * CL/rt_funcs_primitives.cl:113-149).  Same hit-id encoding as above.
 */
void orc_brute_primary_hits( const rt_raytrace_args *a, const void *prims,
	const rt_prim_desc *pd, const rt_camera *cam,
	const rt_triangle *triangles, const rt_vertex *verticies,
	int x0, int xStep, int y0, int yStep,
	int32_t *hitId, float *hitT, int nThreads );

/* single-ray entry points for known-answer tests */
int orc_ray_triangle( const float src[3], const float dest[3],
	const float v0[3], const float v1[3], const float v2[3],
	float *t, float *u, float *v );
int orc_ray_sphere( const float src[3], const float dest[3],
	const float center[3], float rad, float *t );
int orc_ray_box( const rt_box *box, const float src[3], const float dest[3],
	float *tNear, float *tFar );

#ifdef __cplusplus
}
#endif
#endif
