/*
 * POD types shared by the host API, the HIP kernels and the CPU oracle.
 *
 * Binary layout follows the reference's host/device struct ABI
 * (reference rt_types.h:43-200, CL/rt_types.cl:37-163) byte for byte, but
 * without any dependency on <CL/cl.h>: rt_vector3 is a plain 16-byte,
 * 16-byte-aligned {x,y,z,pad} exactly like OpenCL's float3.
 * Every size/offset is pinned by the _Static_asserts at the end of this file.
 */
#ifndef RT_TYPES_H
#define RT_TYPES_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
#define RT_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define RT_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

/* ---- capacity constants (reference rt_types.h:6-12) -------------------- */
#define PRIMS_BLOCK_SIZE    33554432
#define PRIMS_IN_BLOCK      1048576
#define LIGHTS_IN_BLOCK     32
#define LIGHTS_BLOCK_SIZE   1024
#define TRIANGLES_IN_BLOCK  1048576
#define VERTEX_IN_BLOCK     1048576
#define MATERIALS_IN_BLOCK  128

/* ---- kd-tree builder constants (reference rt_types.h:14-17) ------------ *
 * MAX_DEPTH / MAX_PRIMS_IN_NODE are the *defaults* of the run-time
 * parameters RT_PARAM_MAX_DEPTH / RT_PARAM_MAX_PRIMS_IN_NODE.             */
#define SAH_PARTS            32
#define COMPUTE_SAH_THREADS  8
#define MAX_PRIMS_IN_NODE    7
#define MAX_DEPTH            10

/* ---- device constants (reference CL/rt_types.cl:4-15) ------------------ */
#define STACK_SIZE        6      /* rays per pixel: default of RT_PARAM_STACK_SIZE */
#define RT_STACK_SIZE_MAX 8      /* hard upper bound of RT_PARAM_STACK_SIZE        */
#define EPSILON           0.05f  /* secondary-ray origin offset                    */
#define ENV_OPT_DENSITY   1.0f
#define KDTREE_DEPTH      10

typedef enum _RT_PRIMITIVE_TYPE { RT_PT_SPHERE, RT_PT_PLANE, RT_PT_TRIANGLE } RT_PRIMITIVE_TYPE;
typedef enum _RT_AXIS           { RT_AXIS_X, RT_AXIS_Y, RT_AXIS_Z }           RT_AXIS;
typedef enum _RT_LIGHT_TYPE     { RT_LT_POINT, RT_LT_DIRECTED }               RT_LIGHT_TYPE;

typedef float    rt_float;
typedef uint64_t rt_ulong;
typedef int32_t  rt_int;

/* row-vector convention: v' = v * M, translation lives in _41.._43 */
typedef struct _rt_matrix4
{
	rt_float _11, _12, _13, _14;
	rt_float _21, _22, _23, _24;
	rt_float _31, _32, _33, _34;
	rt_float _41, _42, _43, _44;
} rt_matrix4;

typedef rt_matrix4 rt_matrix3;

typedef struct _rt_vector2 { rt_float x, y; } __attribute__((aligned(8)))  rt_vector2;
typedef struct _rt_vector3 { rt_float x, y, z, _pad; } __attribute__((aligned(16))) rt_vector3;
typedef struct _rt_vector4 { rt_float x, y, z, w; } __attribute__((aligned(16))) rt_vector4;

typedef struct _rt_vertex   { rt_vector3 pos; rt_vector3 norm; } rt_vertex;
typedef struct _rt_camera   { rt_matrix4 world; rt_matrix4 viewToPersp; } rt_camera;
typedef struct _rt_argb     { unsigned char b, g, r, a; } rt_argb;
typedef struct _rt_color    { float b, g, r, a; } rt_color;          /* BGRA order */

typedef struct _rt_material
{
	rt_color color;      /* base colour; color.a = opacity (< 0.99 refracts) */
	rt_color ambient;
	rt_color diffuse;
	rt_color specular;   /* specular.a = Phong exponent */
	rt_color reflect;
	rt_float lightFalloff;
	rt_float optDens;
} rt_material;

typedef struct _rt_sphere   { rt_vector3 pos; rt_float rad; rt_ulong mat; } rt_sphere;
typedef struct _rt_plane    { rt_vector3 pos; rt_vector3 norm; rt_ulong mat; } rt_plane;
typedef struct _rt_triangle { rt_ulong pV0, pV1, pV2, mat; } rt_triangle;
typedef struct _rt_box      { rt_vector3 center; rt_vector3 extents; } rt_box;   /* half sizes */
typedef struct _rt_point_light { rt_color col; rt_vector3 pos; rt_float rad; } rt_point_light;

typedef struct _rt_raytrace_args
{
	rt_box   boundingBox;
	rt_ulong primsCount, trianglesCount, vertexCount, lightsCount;
	rt_int   w, h, xdelta, ydelta;
} rt_raytrace_args;

typedef struct _rt_prim_desc  { rt_ulong offset; RT_PRIMITIVE_TYPE pt; } rt_prim_desc;
typedef struct _rt_light_desc { rt_ulong offset; RT_LIGHT_TYPE lt; }     rt_light_desc;

/* flat tree in the reference's interchange format (48 B per node) */
typedef struct _rt_cl_kdtree_node
{
	rt_ulong leftNode, rightNode;
	rt_ulong primsCount, prims;
	float    sep;
	RT_AXIS  axis;
	rt_int   isLast;
} rt_cl_kdtree_node;

typedef struct _rt_kdtree_count_info { rt_ulong primsCount, nodesCount; } rt_kdtree_count_info;

/* host pointer tree produced by rt_kdtree_make_childs */
typedef struct _rt_kdtree_node
{
	struct _rt_kdtree_node *leftNode, *rightNode;
	rt_ulong  primsCount;
	rt_ulong *prims;
	float     sep;
	RT_AXIS   axis;
	int       isLast;
} rt_kdtree_node;

struct rt_pipe_impl;   /* host shadow copies + HIP context, private to the library */

/*
 * Caller-allocated render pipe (reference rt_types.h:211-245).  The OpenCL
 * members (context, queue, kernel, eight cl_mem) are replaced by one opaque
 * pointer; the counters, screenData, w and h keep their names and meaning.
 */
typedef struct _rt_render_pipe
{
	struct rt_pipe_impl *impl;

	rt_ulong primBlocks, primDecsBlocks, lightBlocks, lightDecsBlocks;
	rt_ulong triangleBlocks, vertexBlocks, materialBlocks;

	size_t   primsEnd;
	rt_ulong primsCount;
	size_t   lightsEnd;
	rt_ulong lightsCount;
	rt_ulong trianglesCount;
	rt_ulong vertexCount;

	rt_argb *screenData;
	int w;
	int h;
} rt_render_pipe;

/* ---- ABI pins ----------------------------------------------------------- */
RT_STATIC_ASSERT(sizeof(rt_vector3) == 16 && offsetof(rt_vector3, z) == 8, "rt_vector3");
RT_STATIC_ASSERT(sizeof(rt_vertex) == 32 && offsetof(rt_vertex, norm) == 16, "rt_vertex");
RT_STATIC_ASSERT(sizeof(rt_triangle) == 32 && offsetof(rt_triangle, mat) == 24, "rt_triangle");
RT_STATIC_ASSERT(sizeof(rt_box) == 32 && offsetof(rt_box, extents) == 16, "rt_box");
RT_STATIC_ASSERT(sizeof(rt_sphere) == 32 && offsetof(rt_sphere, rad) == 16 && offsetof(rt_sphere, mat) == 24, "rt_sphere");
RT_STATIC_ASSERT(sizeof(rt_plane) == 48 && offsetof(rt_plane, mat) == 32, "rt_plane");
RT_STATIC_ASSERT(sizeof(rt_point_light) == 48 && offsetof(rt_point_light, pos) == 16 && offsetof(rt_point_light, rad) == 32, "rt_point_light");
RT_STATIC_ASSERT(sizeof(rt_color) == 16 && offsetof(rt_color, r) == 8 && offsetof(rt_color, a) == 12, "rt_color");
RT_STATIC_ASSERT(sizeof(rt_material) == 88 && offsetof(rt_material, reflect) == 64 && offsetof(rt_material, lightFalloff) == 80 && offsetof(rt_material, optDens) == 84, "rt_material");
RT_STATIC_ASSERT(sizeof(rt_argb) == 4, "rt_argb");
RT_STATIC_ASSERT(sizeof(rt_camera) == 128 && offsetof(rt_camera, viewToPersp) == 64, "rt_camera");
RT_STATIC_ASSERT(sizeof(rt_raytrace_args) == 80 && offsetof(rt_raytrace_args, primsCount) == 32 && offsetof(rt_raytrace_args, w) == 64 && offsetof(rt_raytrace_args, ydelta) == 76, "rt_raytrace_args");
RT_STATIC_ASSERT(sizeof(rt_prim_desc) == 16 && offsetof(rt_prim_desc, pt) == 8, "rt_prim_desc");
RT_STATIC_ASSERT(sizeof(rt_light_desc) == 16 && offsetof(rt_light_desc, lt) == 8, "rt_light_desc");
RT_STATIC_ASSERT(sizeof(rt_cl_kdtree_node) == 48 && offsetof(rt_cl_kdtree_node, prims) == 24 && offsetof(rt_cl_kdtree_node, sep) == 32 && offsetof(rt_cl_kdtree_node, axis) == 36 && offsetof(rt_cl_kdtree_node, isLast) == 40, "rt_cl_kdtree_node");
RT_STATIC_ASSERT(sizeof(rt_kdtree_node) == 48 && offsetof(rt_kdtree_node, sep) == 32, "rt_kdtree_node");

#endif
