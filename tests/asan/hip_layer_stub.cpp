// TEST INFRASTRUCTURE, not part of the product: a host-only stand-in for the
// rt_hip_* layer (include/rt_hip.h), linked ONLY into the sanitizer build of the
// host sources (make -C raytracing_amd/csrc asan -> librt_amd_asan.so).  It
// lets the render pipe's scene store, block accounting, OBJ loader and kd
// builder run under AddressSanitizer / UndefinedBehaviorSanitizer on a machine
// without a GPU.  It renders nothing: every upload reads all the bytes it is
// given (so the sanitizer checks the caller's buffers) and frames come back
// black.  The product library never contains this file.
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "rt_hip.h"
#include "rt_kdtree.h"

struct rt_hip_ctx { int w = 0, h = 0, bandH = 16, parts = 1, part = 0; uint64_t sum = 0; };

static uint64_t touch( const void *p, size_t bytes )
{
	uint64_t s = 0;
	const unsigned char *b = (const unsigned char *) p;
	for ( size_t i = 0; i < bytes; ++i ) s += b[i];
	return s;
}

extern "C" {

int rt_hip_device_count( void ) { return 1; }
const char *rt_hip_last_error( void ) { return "sanitizer stand-in: no device"; }
rt_hip_ctx *rt_hip_create( int ) { return new rt_hip_ctx(); }
void rt_hip_destroy( rt_hip_ctx *c ) { delete c; }
int rt_hip_set_stream( rt_hip_ctx *, void * ) { return 0; }
int rt_hip_upload_tree( rt_hip_ctx *c, const rt_hip_kdnode *n, uint64_t nn, const rt_hip_trirec *r, uint64_t nr,
	const rt_hip_trishade *s, uint64_t nt, const rt_box *b, int )
{
	c->sum += touch( n, nn * sizeof( *n ) ) + touch( r, nr * sizeof( *r ) ) + touch( s, nt * sizeof( *s ) ) + touch( b, sizeof( *b ) );
	return 0;
}
int rt_hip_upload_tree_lists( rt_hip_ctx *c, const rt_hip_kdnode *n, uint64_t nn, const rt_hip_trirec *r, uint64_t nr, uint64_t nf,
	const rt_hip_leaf_full *lf, const rt_hip_trishade *s, uint64_t nt, const rt_box *b, int )
{
	c->sum += touch( n, nn * sizeof( *n ) ) + touch( r, ( nr + ( lf ? nf : 0 ) ) * sizeof( *r ) ) + touch( s, nt * sizeof( *s ) ) + touch( b, sizeof( *b ) );
	if ( lf ) c->sum += touch( lf, nn * sizeof( *lf ) );
	return 0;
}
int rt_hip_adopt_tree( rt_hip_ctx *, struct _rt_kdtree_device * ) { return 1; }
int rt_hip_build_tree( rt_hip_ctx *c, const rt_vertex *v, const rt_triangle *t, uint64_t nv, uint64_t nt, int, int )
{
	c->sum += touch( v, nv * sizeof( *v ) ) + touch( t, nt * sizeof( *t ) );
	return 0;
}
int rt_hip_stage_geometry( rt_hip_ctx *c, const rt_vertex *v, uint64_t nv, const rt_triangle *t, uint64_t nt )
{
	c->sum += touch( v, nv * sizeof( *v ) ) + touch( t, nt * sizeof( *t ) );
	return 0;
}
void rt_hip_unstage_geometry( rt_hip_ctx * ) {}
int rt_hip_upload_materials( rt_hip_ctx *c, const rt_material *m, uint64_t n ) { c->sum += touch( m, n * sizeof( *m ) ); return 0; }
int rt_hip_upload_spheres( rt_hip_ctx *c, const rt_sphere *s, uint64_t n ) { c->sum += touch( s, n * sizeof( *s ) ); return 0; }
int rt_hip_upload_lights( rt_hip_ctx *c, const rt_point_light *l, uint64_t n ) { c->sum += touch( l, n * sizeof( *l ) ); return 0; }
int rt_hip_share_scene( rt_hip_ctx *, const rt_hip_ctx * ) { return 0; }
int rt_hip_set_camera( rt_hip_ctx *c, const rt_camera *cam ) { c->sum += touch( cam, sizeof( *cam ) ); return 0; }
int rt_hip_set_image( rt_hip_ctx *c, int w, int h ) { c->w = w; c->h = h; return w < 2 || h < 2; }
int rt_hip_set_tile( rt_hip_ctx *c, int bandH, int parts, int part ) { c->bandH = bandH; c->parts = parts; c->part = part; return 0; }
int rt_hip_local_rows( rt_hip_ctx *c )
{
	int rows = 0;
	for ( int b = c->part; b * c->bandH < c->h; b += c->parts )
		rows += c->h - b * c->bandH < c->bandH ? c->h - b * c->bandH : c->bandH;
	return rows;
}
int rt_hip_set_param( rt_hip_ctx *, int, int ) { return 0; }
int rt_hip_render( rt_hip_ctx *, void *, unsigned ) { return 0; }
int rt_hip_render_noalloc( rt_hip_ctx *, void *, unsigned ) { return 1; }
int rt_hip_sync( rt_hip_ctx * ) { return 0; }
int rt_hip_read_frame( rt_hip_ctx *c, rt_argb *host ) { memset( host, 0, sizeof( rt_argb ) * (size_t) c->w * rt_hip_local_rows( c ) ); return 0; }
int rt_hip_read_hit_ids( rt_hip_ctx *, int32_t * ) { return 1; }
void *rt_hip_alloc_host( rt_hip_ctx *, size_t ) { return NULL; }          // callers fall back to malloc, which the sanitizer sees
void rt_hip_free_host( rt_hip_ctx *, void * ) {}
int rt_hip_get_stats( rt_hip_ctx *, rt_hip_stats * ) { return 1; }
int rt_hip_kernel_times( rt_hip_ctx *, float *, int * ) { return 1; }
int rt_hip_kernel_times_ex( rt_hip_ctx *, float *, int *, float * ) { return 1; }
int rt_hip_kernel_intervals( rt_hip_ctx *, rt_hip_ctx *, float *, int *, int, int * ) { return 1; }
int rt_hip_selftest_rcp( rt_hip_ctx *, uint64_t *, uint64_t * ) { return 1; }

// GPU kd builder (kd_build_gpu.hip): not available in this build
int rt_kdtree_build_device( const rt_vertex *, const rt_triangle *, uint64_t, uint64_t, int, int, void *, rt_kdtree_device * ) { return 1; }
int rt_kdtree_build_device_ex( const rt_vertex *, const rt_triangle *, uint64_t, uint64_t, const rt_kdtree_build_opts *, void *, rt_kdtree_device * ) { return 1; }
void rt_kdtree_device_release( rt_kdtree_device * ) {}
int rt_kdtree_device_download( const rt_kdtree_device *, rt_kdtree_flat * ) { return 1; }
const char *rt_kdtree_device_last_error( void ) { return "sanitizer stand-in: no device"; }
void rt_kdtree_device_trim( void ) {}

}
