/*
 * The HIP layer of the library (include/rt_hip.h, include/rt_kdtree.h) used
 * from plain C, without the render pipe: build a flat kd-tree on the host,
 * upload the scene, render into the context's own frame buffer, read the
 * frame back, then render the same scene from a second context that shares
 * the first one's resident scene (frames in flight), with two frames enqueued
 * before either is read.
 *
 *   make -C raytracing_amd/csrc app && apps/abi_example [w h]
 *
 * Prints the FNV-1a hash of the frame (tests/test_parity_gpu.py renders the
 * same scene through the Python mirror and compares) and exits non-zero on
 * any error or if the two contexts disagree.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rt.h"
#include "rt_hip.h"
#include "rt_kdtree.h"

#define N 24                      /* the scene: an N x N grid of quads with a bump, a sphere, one light */

static int die( const char *what )
{
	fprintf( stderr, "abi_example: %s: %s\n", what, rt_hip_last_error() );
	return 1;
}

static uint64_t fnv( const void *p, size_t n )
{
	const unsigned char *b = (const unsigned char *) p;
	uint64_t h = 1469598103934665603ull;
	size_t i;
	for ( i = 0; i < n; ++i ) { h ^= b[i]; h *= 1099511628211ull; }
	return h;
}

int main( int argc, char **argv )
{
	const int w = argc > 2 ? atoi( argv[1] ) : 160, h = argc > 2 ? atoi( argv[2] ) : 96;
	const int nv = ( N + 1 ) * ( N + 1 ), nt = 2 * N * N;
	rt_vertex *v = (rt_vertex *) calloc( nv, sizeof( rt_vertex ) );
	rt_triangle *t = (rt_triangle *) calloc( nt, sizeof( rt_triangle ) );
	rt_argb *f0 = (rt_argb *) malloc( sizeof( rt_argb ) * w * h ), *f1 = (rt_argb *) malloc( sizeof( rt_argb ) * w * h );
	rt_material mat[2];
	rt_color c0, c1, c2, c3, c4;
	rt_vector3 p;
	rt_sphere sph;
	rt_point_light light;
	rt_camera cam;
	rt_kdtree_flat tree;
	rt_hip_ctx *a, *b;
	int i, j, k = 0;

	if ( !v || !t || !f0 || !f1 || w < 8 || h < 8 ) return 1;
	/* a floor below the camera axis (+y is down the image) with one smooth bump, facing up */
	for ( j = 0; j <= N; ++j )
		for ( i = 0; i <= N; ++i )
		{
			const float x = -60.0f + 120.0f * (float) i / N, z = 40.0f + 120.0f * (float) j / N;
			const float dx = x / 30.0f, dz = ( z - 100.0f ) / 30.0f;
			rt_vertex *q = &v[j * ( N + 1 ) + i];
			rt_vector3_create( &q->pos, x, 25.0f - 12.0f / ( 1.0f + dx * dx + dz * dz ), z );
			rt_vector3_create( &q->norm, 0.0f, -1.0f, 0.0f );
		}
	for ( j = 0; j < N; ++j )
		for ( i = 0; i < N; ++i )
		{
			const rt_ulong o = (rt_ulong) ( j * ( N + 1 ) + i );
			t[k].pV0 = o; t[k].pV1 = o + 1; t[k].pV2 = o + N + 1; t[k].mat = 0; ++k;
			t[k].pV0 = o + 1; t[k].pV1 = o + N + 2; t[k].pV2 = o + N + 1; t[k].mat = 0; ++k;
		}
	rt_color_create( &c0, 1.0f, 0.8f, 0.7f, 0.5f );           /* a, r, g, b */
	rt_color_create( &c1, 0.0f, 0.05f, 0.05f, 0.05f );
	rt_color_create( &c2, 0.0f, 0.6f, 0.6f, 0.6f );
	rt_color_create( &c3, 20.0f, 0.3f, 0.3f, 0.3f );
	rt_color_create( &c4, 0.0f, 0.4f, 0.4f, 0.4f );
	rt_material_create( &mat[0], &c0, &c1, &c2, &c3, &c4, 0.0f, 1.0f );
	rt_color_create( &c0, 1.0f, 0.2f, 0.4f, 0.9f );
	rt_material_create( &mat[1], &c0, &c1, &c2, &c3, &c4, 0.0f, 1.0f );
	rt_vector3_create( &p, 10.0f, 0.0f, 90.0f );
	rt_sphere_create( &sph, &p, 9.0f, 1 );
	rt_color_create( &c0, 1.0f, 1.0f, 1.0f, 1.0f );
	rt_vector3_create( &p, -20.0f, -35.0f, 0.0f );
	rt_point_light_create( &light, &p, 100.0f, &c0 );
	rt_camera_create( &cam, (float) w / (float) h, 0.125f * 3.14159265f );

	if ( rt_hip_device_count() < 1 ) return die( "no HIP device" );
	if ( rt_kdtree_build_flat( v, t, (uint64_t) nv, (uint64_t) nt, 12, 4, 1, &tree ) ) { fprintf( stderr, "abi_example: tree build failed\n" ); return 1; }
	if ( !( a = rt_hip_create( 0 ) ) || !( b = rt_hip_create( 0 ) ) ) return die( "rt_hip_create" );
	if ( rt_hip_upload_tree( a, tree.nodes, tree.nNodes, tree.recs, tree.nRecs, tree.shade, tree.nTris, &tree.bbox, tree.depth ) ) return die( "rt_hip_upload_tree" );
	if ( rt_hip_upload_materials( a, mat, 2 ) || rt_hip_upload_spheres( a, &sph, 1 ) || rt_hip_upload_lights( a, &light, 1 ) ) return die( "upload" );
	if ( rt_hip_set_camera( a, &cam ) || rt_hip_set_image( a, w, h ) || rt_hip_set_param( a, RT_PARAM_STACK_SIZE, 4 ) ) return die( "set up" );
	/* the second context reads the first one's scene */
	if ( rt_hip_share_scene( b, a ) ) return die( "rt_hip_share_scene" );
	if ( rt_hip_set_camera( b, &cam ) || rt_hip_set_image( b, w, h ) || rt_hip_set_param( b, RT_PARAM_STACK_SIZE, 4 ) ) return die( "set up (second context)" );

	/* frames in flight: a pass of either frame takes half the wave slots of a compute unit, so that the other frame's
	 * passes find room beside it (the default, all 16, is for one frame at a time) */
	if ( rt_hip_set_param( a, RT_PARAM_LAUNCH_WAVES, 8 ) || rt_hip_set_param( b, RT_PARAM_LAUNCH_WAVES, 8 ) ) return die( "RT_PARAM_LAUNCH_WAVES" );
	/* two frames in flight: both are enqueued before either is read */
	if ( rt_hip_render( a, NULL, 0 ) || rt_hip_render( b, NULL, 0 ) ) return die( "rt_hip_render" );
	if ( rt_hip_read_frame( a, f0 ) || rt_hip_read_frame( b, f1 ) ) return die( "rt_hip_read_frame" );
	if ( memcmp( f0, f1, sizeof( rt_argb ) * w * h ) ) { fprintf( stderr, "abi_example: the two contexts rendered different frames\n" ); return 2; }
	for ( i = 0, k = 0; i < w * h; ++i ) k += ( f0[i].r | f0[i].g | f0[i].b ) != 0;
	if ( k < w * h / 8 ) { fprintf( stderr, "abi_example: the frame is almost empty (%d lit pixels)\n", k ); return 3; }
	printf( "%dx%d triangles %d nodes %llu lit %d fnv %016llx\n", w, h, nt, (unsigned long long) tree.nNodes, k,
		(unsigned long long) fnv( f0, sizeof( rt_argb ) * w * h ) );

	rt_hip_destroy( b );
	rt_hip_destroy( a );
	rt_kdtree_flat_release( &tree );
	free( v ); free( t ); free( f0 ); free( f1 );
	return 0;
}
