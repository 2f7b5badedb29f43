// Constructors behind include/rt_funcs_primitives.h (reference
// rt_funcs_primitives.c:3-85): plain field copies, exit(-1) on NULL.
#include <cstdlib>

#include "rt_funcs_primitives.h"

static void need( const void *a, const void *b = (const void *) 1, const void *c = (const void *) 1 )
{
	if ( !a || !b || !c )
		exit( -1 );
}

extern "C" {

void rt_sphere_create( rt_sphere *pS, rt_vector3 *pP, rt_float r, unsigned long int mat )
{
	need( pS, pP );
	pS->pos = *pP;
	pS->rad = r;
	pS->mat = mat;
}

void rt_plane_create( rt_plane *pPl, rt_vector3 *pP, rt_vector3 *pN, unsigned long int mat )
{
	need( pPl, pP, pN );
	pPl->pos = *pP;
	pPl->norm = *pN;
	pPl->mat = mat;
}

void rt_material_create( rt_material *pM, rt_color *pC, rt_color *pA, rt_color *pD, rt_color *pS, rt_color *pR,
	rt_float rFoff, rt_float oD )
{
	need( pM, pC, pA );
	need( pD, pS, pR );
	pM->color = *pC;
	pM->ambient = *pA;
	pM->diffuse = *pD;
	pM->specular = *pS;
	pM->reflect = *pR;
	pM->lightFalloff = rFoff;
	pM->optDens = oD;
}

void rt_point_light_create( rt_point_light *pS, rt_vector3 *pP, rt_float r, rt_color *pCol )
{
	need( pS, pP, pCol );
	pS->pos = *pP;
	pS->col = *pCol;
	pS->rad = r;
}

void rt_camera_create( rt_camera *pC, rt_float asp, rt_float hV )
{
	need( pC );
	rt_matrix4_create_translate( &pC->world, 0.0f, 0.0f, 0.0f );
	rt_matrix4_create_projection( &pC->viewToPersp, asp, hV );
}

void rt_camera_set_view_matrix( rt_camera *pC, rt_matrix4 *pM )
{
	need( pC, pM );
	pC->world = *pM;
}

void rt_camera_set_perspective_matrix( rt_camera *pC, rt_matrix4 *pM )
{
	need( pC, pM );
	pC->viewToPersp = *pM;
}

} // extern "C"
