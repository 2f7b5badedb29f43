/*
 * Triangle-mesh container and loader: the API of the reference's
 * rt_meshes.h:6-33.  rt_mesh_add_group takes (pipe, mesh, material slot,
 * group index), the order of the reference's definition
 * (rt_meshes.c:352-353), which is what its callers rely on.
 */
#ifndef RT_MESHES_H
#define RT_MESHES_H

#include "rt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct _rt_mesh_group
{
	rt_material *mat;
	rt_ulong voffset, toffset;
	rt_ulong vc, tc;
} rt_mesh_group;

typedef struct _rt_mesh
{
	rt_vertex *v;
	rt_triangle *t;
	rt_mesh_group *g;
	rt_material *mat;
	rt_ulong mc, gc, vc, tc;
} rt_mesh;

/* Wavefront OBJ (+ MTL) loader; returns -1 if the file cannot be opened */
int  rt_mesh_load_from_obj( rt_mesh *m, char *path );
/* recompute vertex normals from the faces (rt_meshes.c:77-107) */
void rt_restore_normals( rt_mesh *m );
void rt_mesh_add( rt_render_pipe *pRp, rt_mesh *m, rt_ulong tg );
void rt_mesh_add_group( rt_render_pipe *pRp, rt_mesh *m, rt_ulong tg, rt_ulong g );
void rt_mesh_reserve( rt_mesh *m, rt_ulong vc, rt_ulong tc, rt_ulong gc );
void rt_mesh_release( rt_mesh *m );

#ifdef __cplusplus
}
#endif
#endif
