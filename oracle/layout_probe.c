/*
 * Test infrastructure.  Compiled against the REFERENCE's rt_types.h (include
 * path set by oracle/Makefile, target `ref`) and prints the size and member
 * offsets of every struct on the hot path as JSON.  The output is committed
 * as tests/golden/ref_layout.json and compared with this repo's own
 * include/rt_types.h by the ABI test.
 */
#include <stddef.h>
#include <stdio.h>
#include "rt_types.h"

#define S(T) printf( "%s\"%s\": {\"size\": %zu", first ? "" : ",\n", #T, sizeof(T) ); first = 0;
#define O(T, m) printf( ", \"%s\": %zu", #m, offsetof(T, m) );
#define E printf( "}" );

int main( void )
{
	int first = 1;
	printf( "{\n" );
	S(rt_vector3) E
	S(rt_vertex) O(rt_vertex, pos) O(rt_vertex, norm) E
	S(rt_triangle) O(rt_triangle, pV0) O(rt_triangle, pV1) O(rt_triangle, pV2) O(rt_triangle, mat) E
	S(rt_box) O(rt_box, center) O(rt_box, extents) E
	S(rt_sphere) O(rt_sphere, pos) O(rt_sphere, rad) O(rt_sphere, mat) E
	S(rt_plane) O(rt_plane, pos) O(rt_plane, norm) O(rt_plane, mat) E
	S(rt_point_light) O(rt_point_light, col) O(rt_point_light, pos) O(rt_point_light, rad) E
	S(rt_color) O(rt_color, b) O(rt_color, g) O(rt_color, r) O(rt_color, a) E
	S(rt_material) O(rt_material, color) O(rt_material, ambient) O(rt_material, diffuse)
		O(rt_material, specular) O(rt_material, reflect) O(rt_material, lightFalloff) O(rt_material, optDens) E
	S(rt_argb) E
	S(rt_matrix4) O(rt_matrix4, _11) O(rt_matrix4, _21) O(rt_matrix4, _41) O(rt_matrix4, _44) E
	S(rt_camera) O(rt_camera, world) O(rt_camera, viewToPersp) E
	S(rt_raytrace_args) O(rt_raytrace_args, boundingBox) O(rt_raytrace_args, primsCount)
		O(rt_raytrace_args, trianglesCount) O(rt_raytrace_args, vertexCount) O(rt_raytrace_args, lightsCount)
		O(rt_raytrace_args, w) O(rt_raytrace_args, h) O(rt_raytrace_args, xdelta) O(rt_raytrace_args, ydelta) E
	S(rt_prim_desc) O(rt_prim_desc, offset) O(rt_prim_desc, pt) E
	S(rt_light_desc) O(rt_light_desc, offset) O(rt_light_desc, lt) E
	S(rt_cl_kdtree_node) O(rt_cl_kdtree_node, leftNode) O(rt_cl_kdtree_node, rightNode)
		O(rt_cl_kdtree_node, primsCount) O(rt_cl_kdtree_node, prims) O(rt_cl_kdtree_node, sep)
		O(rt_cl_kdtree_node, axis) O(rt_cl_kdtree_node, isLast) E
	S(rt_kdtree_node) O(rt_kdtree_node, leftNode) O(rt_kdtree_node, rightNode) O(rt_kdtree_node, primsCount)
		O(rt_kdtree_node, prims) O(rt_kdtree_node, sep) O(rt_kdtree_node, axis) O(rt_kdtree_node, isLast) E
	S(rt_kdtree_count_info) O(rt_kdtree_count_info, primsCount) O(rt_kdtree_count_info, nodesCount) E
	printf( "\n}\n" );
	return 0;
}
