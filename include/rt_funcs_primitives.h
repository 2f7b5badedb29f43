/*
 * Constructors for spheres, planes, materials, lights and the camera: the
 * API of the reference's rt_funcs_primitives.h:11-28.  All of them copy
 * their arguments; a NULL argument terminates the process with exit(-1)
 * like the reference does.
 */
#ifndef RT_FUNCS_PRIMITIVES_H
#define RT_FUNCS_PRIMITIVES_H

#include "rt_types.h"
#include "rt_funcs_math.h"

#ifdef __cplusplus
extern "C" {
#endif

void rt_sphere_create( rt_sphere *pS, rt_vector3 *pP, rt_float r, unsigned long int mat );
void rt_plane_create( rt_plane *pPl, rt_vector3 *pP, rt_vector3 *pN, unsigned long int mat );
void rt_material_create( rt_material *pM, rt_color *pC, rt_color *pA, rt_color *pD, rt_color *pS,
	rt_color *pR, rt_float rFoff, rt_float oD );
void rt_point_light_create( rt_point_light *pS, rt_vector3 *pP, rt_float r, rt_color *pCol );
/* identity world matrix + rt_matrix4_create_projection( asp, hV ) */
void rt_camera_create( rt_camera *pC, rt_float asp, rt_float hV );
void rt_camera_set_view_matrix( rt_camera *pC, rt_matrix4 *pM );
void rt_camera_set_perspective_matrix( rt_camera *pC, rt_matrix4 *pM );

#ifdef __cplusplus
}
#endif
#endif
