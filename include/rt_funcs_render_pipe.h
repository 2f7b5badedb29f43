/*
 * Render-pipe API: scene load + render.  Same function names, argument
 * meaning, ownership and error behaviour as the reference's
 * rt_funcs_render_pipe.h:15-70, so a caller written against the reference
 * (its main.c) compiles and runs unchanged; underneath, the OpenCL
 * context/queue/program plumbing is replaced by the HIP layer of rt_hip.h.
 *
 * Contract kept from the reference:
 *   - rt_render_pipe is caller-allocated, set up by _create, torn down by
 *     _cleanup;  every add_* call copies its argument;
 *   - rt_render_pipe_draw builds the kd-tree over everything added since the
 *     previous draw, renders, returns pRp->screenData (owned by the pipe,
 *     w*h BGRA bytes, row 0 first) and resets triangles / vertices / spheres
 *     / lights, while materials and the camera persist
 *     (reference rt_funcs_render_pipe.c:1211-1234);
 *   - NULL arguments exit(-1); device / runtime failures print a message and
 *     exit(-3) (reference :23-24, :450-492).
 *
 * The functions that took cl_mem / OpenCL handles (rt_init_opencl,
 * rt_init_buffers, rt_add_block, rt_opencl_render, rt_render_pipe_calc_wg_size)
 * have HIP-typed successors in rt_hip.h; rt_init_buffers and rt_add_block
 * keep their host-visible effect (block counters) here.
 */
#ifndef RT_FUNCS_RENDER_PIPE_H
#define RT_FUNCS_RENDER_PIPE_H

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rt_types.h"
#include "rt_funcs_math.h"
#include "rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* directory of the executable; the reference located its kernel sources with
 * these (rt_funcs_render_pipe.h:12-13).  The HIP kernels are linked into the
 * library, the strings are kept for source compatibility only. */
extern char *rt_cl_include_path;
extern char *rt_cl_raytrace_kernel_path;

void rt_init( char *path );

void rt_render_pipe_create( rt_render_pipe *pRp, int w, int h );
void rt_render_pipe_set_image_size( rt_render_pipe *pRp, int w, int h );
void rt_render_pipe_reset_blocks( rt_render_pipe *pRp );
void rt_init_buffers( rt_render_pipe *pRp );

void rt_render_pipe_set_camera( rt_render_pipe *pRp, rt_camera *pFr );
/* lends a pointer to the pipe's camera; edits take effect at the next draw */
void rt_render_pipe_get_camera( rt_render_pipe *pRp, rt_camera **ppC );
void rt_render_pipe_free_camera( rt_render_pipe *pRp, rt_camera *pC );

void rt_render_pipe_add_primitive( rt_render_pipe *pRp, void *pPrim, RT_PRIMITIVE_TYPE pt );
/* appends vCount vertices from pV + vOffset and tCount triangles from
 * pTr + tOffset; vertex indices are rebased to the pipe's global vertex
 * array and every triangle gets material slot tGroup (reference :261-305) */
void rt_render_pipe_add_triangles( rt_render_pipe *pRp, rt_vertex *pV, rt_triangle *pTr,
	rt_ulong vOffset, rt_ulong tOffset, rt_ulong vCount, rt_ulong tCount, rt_ulong tGroup );
void rt_render_pipe_add_material( rt_render_pipe *pRp, rt_material *mat, unsigned long int ind );
void rt_render_pipe_add_light( rt_render_pipe *pRp, void *pLight, RT_LIGHT_TYPE lt );

/* host kd-tree builder, identical results to the reference's
 * (rt_funcs_render_pipe.c:591-1089): pointer tree, SAH split, flat packing */
rt_kdtree_count_info rt_kdtree_make_childs( rt_vertex *pV, rt_triangle *pTr, rt_kdtree_node *pNode,
	rt_box *bbox, int depth );
void rt_kdtree_compute_sah( rt_ulong *memTp, rt_triangle *memTr, rt_vertex *memVer,
	rt_ulong primsCount, RT_AXIS axis, rt_box *pBox, rt_ulong *primsCountL, rt_ulong *primsCountR,
	float *sep );
rt_kdtree_count_info rt_kdtree_pack_to_buffer( rt_cl_kdtree_node *pNodeBuf, rt_ulong *pPrimsIdxBuf,
	rt_kdtree_node *pNode, rt_ulong writePos, rt_ulong primsWritePos );
/* Builds the tree over the pipe's current geometry and returns it in the
 * reference's flat interchange format.  The reference returned two cl_mem
 * (:1091-1209); here *nodes / *primsIdx are malloc'ed host arrays the caller
 * frees.  rt_render_pipe_draw itself uses the HBM layout of rt_kdtree.h. */
void rt_kdtree_build( rt_render_pipe *pRp, rt_box *pBoundingBox, rt_cl_kdtree_node **nodes,
	rt_ulong *nodesCount, rt_ulong **primsIdx, rt_ulong *primsIdxCount );

rt_argb *rt_render_pipe_draw( rt_render_pipe *pRp );
void rt_render_pipe_cleanup( rt_render_pipe *pRp );

/* ---- additions without a reference counterpart ------------------------- */

/* RT_PARAM_* keys of rt_hip.h; the reference fixes these at compile time */
void rt_render_pipe_set_param( rt_render_pipe *pRp, int key, int value );
/* multi-GPU split: this pipe renders bands part, part+nParts, ... of bandH rows */
void rt_render_pipe_set_tile( rt_render_pipe *pRp, int bandH, int nParts, int part );
/* HIP device ordinal used by pipes created afterwards (default 0) */
void rt_render_pipe_select_device( int device );
/* the pipe's HIP context, for callers that keep the scene resident */
rt_hip_ctx *rt_render_pipe_hip( rt_render_pipe *pRp );
/* like rt_render_pipe_draw but leaves the frame in device memory d_out
 * (w * local rows rt_argb) and does not wait for it */
void rt_render_pipe_draw_device( rt_render_pipe *pRp, void *d_out );

#ifdef __cplusplus
}
#endif
#endif
