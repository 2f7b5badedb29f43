/*
 * Headless frame renderer on the reference's C API (rt.h, rt_meshes.h) linked
 * against librt_amd.so: what the reference's GTK application does per frame
 * (main.c:656-754: add materials, light, spheres, box, water, draw; main.c:69-86:
 * advance t), without a window.  Frames go to <out>/<n>.png (or .ppm), the
 * numbering and the `save` file (t, dT, next frame number; main.c:88-100,
 * 793-800) follow the reference, so its build_video.sh works on the output
 * and an interrupted run resumes where it stopped.
 *
 * The scene is the reference's demo scene (water surface of seven sine waves
 * in an open box, one large and nine orbiting spheres, one point light); the
 * pseudo-random sphere colours and orbit factors come from a fixed generator
 * instead of rand() so that runs are reproducible.
 *
 *   rt_headless [--frames N] [--size WxH] [--t T] [--dt DT] [--out DIR]
 *               [--format png|ppm|none] [--resume] [--depth D] [--leaf L]
 *               [--stack S] [--squares N] [--obj FILE] [--device N] [--gpu-build | --host-build] [--prune-leaves | --reference-leaves]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <zlib.h>

#include "rt.h"
#include "rt_meshes.h"

#define N_ORBIT 9

static unsigned int g_seed = 12345u;
static unsigned int demo_rand( void )
{
	g_seed = ( g_seed * 1103515245u + 12345u ) & 0x7fffffffu;
	return g_seed >> 8;
}

/* period, amplitude, angle, speed of the water waves (main.c:330-349) */
static const float g_waves[7][4] = {
	{ 100.0f, 1.5f, 1.57079633f, 500.0f }, { 79.16677f, 0.25f, 2.9428474f, 256.9786f },
	{ 144.04169f, 0.25f, -1.8729727f, 160.6270f }, { 79.15634f, 0.25f, 1.8583484f, 239.9041f },
	{ 121.51403f, 0.5f, -1.5343947f, 230.2120f }, { 72.40281f, 0.5f, 1.4335813f, 300.9681f },
	{ 81.57271f, 0.5f, -1.2057098f, 280.5781f } };

typedef struct
{
	rt_material mats[12];
	float koef[1 + N_ORBIT];
	rt_point_light light;
	rt_mesh box, water;
	int squares;
	float plSz, plHt;
} demo_scene;

static void material( rt_material *m, float ca, float cr, float cg, float cb, float amb, float dif,
	float spa, float spc, float refl, float falloff, float dens )
{
	rt_color c, a, d, s, r;
	rt_color_create( &c, ca, cr, cg, cb );
	rt_color_create( &a, 0.0f, amb, amb, amb );
	rt_color_create( &d, 0.0f, dif, dif, dif );
	rt_color_create( &s, spa, spc, spc, spc );
	rt_color_create( &r, 0.0f, refl, refl, refl );
	rt_material_create( m, &c, &a, &d, &s, &r, falloff, dens );
}

static void set_vertex( rt_vertex *v, float x, float y, float z, float nx, float ny, float nz )
{
	rt_vector3_create( &v->pos, x, y, z );
	rt_vector3_create( &v->norm, nx, ny, nz );
}

static void set_tri( rt_triangle *t, rt_ulong a, rt_ulong b, rt_ulong c )
{
	t->pV0 = a; t->pV1 = b; t->pV2 = c; t->mat = 0;
}

/* open box around the water: bottom and four walls, normals pointing inwards (main.c:465-611) */
static void build_box( demo_scene *sc )
{
	const float hx = sc->plSz * 0.5f, z0 = -sc->plSz * 0.5f + 50.0f, z1 = sc->plSz * 0.5f + 50.0f;
	const float top = sc->plHt - 3.0f, bot = sc->plHt - 3.0f + 25.0f;
	const float p[20][3] = {
		{ hx, bot, z1 }, { hx, bot, z0 }, { -hx, bot, z1 }, { -hx, bot, z0 },
		{ -hx, bot, z1 }, { -hx, bot, z0 }, { -hx, top, z1 }, { -hx, top, z0 },
		{ hx, bot, z1 }, { hx, bot, z0 }, { hx, top, z1 }, { hx, top, z0 },
		{ -hx, bot, z1 }, { hx, bot, z1 }, { -hx, top, z1 }, { hx, top, z1 },
		{ -hx, bot, z0 }, { hx, bot, z0 }, { -hx, top, z0 }, { hx, top, z0 } };
	const float n[5][3] = { { 0, -1, 0 }, { 1, 0, 0 }, { -1, 0, 0 }, { 0, 0, -1 }, { 0, 0, 1 } };
	const int t[10][3] = { { 0, 1, 2 }, { 3, 1, 2 }, { 4, 5, 6 }, { 7, 6, 5 }, { 8, 9, 10 }, { 11, 10, 9 },
		{ 12, 13, 14 }, { 15, 14, 13 }, { 16, 17, 18 }, { 19, 18, 17 } };
	int i;
	memset( &sc->box, 0, sizeof( sc->box ) );
	rt_mesh_reserve( &sc->box, 20, 10, 1 );
	for ( i = 0; i < 20; ++i )
		set_vertex( &sc->box.v[i], p[i][0], p[i][1], p[i][2], n[i / 4][0], n[i / 4][1], n[i / 4][2] );
	for ( i = 0; i < 10; ++i )
		set_tri( &sc->box.t[i], t[i][0], t[i][1], t[i][2] );
}

/* water surface at time t: heights, then normals = (0,-1,0) + sum of the
 * adjacent face normals, normalised (main.c:351-423) */
static void build_water( demo_scene *sc, float t )
{
	const int pr = sc->squares + 1;
	const float dt = sc->plSz / (float) sc->squares;
	int i, j, k;
	rt_ulong q = 0;
	for ( i = 0; i < pr; ++i )
		for ( j = 0; j < pr; ++j )
		{
			const float u = (float) i / (float) pr, v = (float) j / (float) pr;
			float y = sc->plHt;
			for ( k = 0; k < 7; ++k )
				y = (float) ( (double) y + sin( (double) g_waves[k][0] * ( -cos( (double) g_waves[k][2] ) * u
					+ sin( (double) g_waves[k][2] ) * v ) + (double) (float) ( t * g_waves[k][3] ) ) * g_waves[k][1] );
			set_vertex( &sc->water.v[i * pr + j], -sc->plSz * 0.5f + (float) i * dt, y,
				50.0f - sc->plSz * 0.5f + (float) j * dt, 0.0f, -1.0f, 0.0f );
		}
	for ( i = 0; i < sc->squares; ++i )
		for ( j = 0; j < sc->squares; ++j )
		{
			set_tri( &sc->water.t[q++], pr * i + j, pr * ( i + 1 ) + j, pr * i + j + 1 );
			set_tri( &sc->water.t[q++], pr * ( i + 1 ) + j + 1, pr * i + j + 1, pr * ( i + 1 ) + j );
		}
	/* triangle by triangle, each normal added to its three corners in turn: the
	 * per-vertex fp32 summation order of the reference (main.c:398-418) */
	for ( q = 0; q < sc->water.tc; ++q )
	{
		rt_triangle *tr = &sc->water.t[q];
		rt_vertex *a = &sc->water.v[tr->pV0], *b = &sc->water.v[tr->pV1], *c = &sc->water.v[tr->pV2];
		const float e1[3] = { b->pos.x - a->pos.x, b->pos.y - a->pos.y, b->pos.z - a->pos.z };
		const float e2[3] = { c->pos.x - a->pos.x, c->pos.y - a->pos.y, c->pos.z - a->pos.z };
		const float n[3] = { e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0] };
		rt_vertex *corner[3] = { a, b, c };
		for ( k = 0; k < 3; ++k )
		{
			corner[k]->norm.x += n[0]; corner[k]->norm.y += n[1]; corner[k]->norm.z += n[2];
		}
	}
	for ( q = 0; q < sc->water.vc; ++q )
	{
		rt_vector3 *n = &sc->water.v[q].norm;
		const float l = sqrtf( n->x * n->x + n->y * n->y + n->z * n->z );
		n->x /= l; n->y /= l; n->z /= l;
	}
}

static void demo_init( demo_scene *sc, int squares )
{
	int i;
	rt_vector3 lp;
	rt_color lc;
	memset( sc, 0, sizeof( *sc ) );
	sc->plSz = 150.0f;
	sc->plHt = 20.0f;
	sc->squares = squares;
	material( &sc->mats[0], 0.5f, 0.223f, 0.345f, 0.474f, 0.05f, 0.5f, 3.5f, 0.25f, 0.3f, 0.25f, 1.33f );   /* water */
	material( &sc->mats[1], 0.5f, 0.0f, 0.5f, 1.0f, 0.05f, 0.2f, 15.0f, 0.2f, 0.5f, 0.0f, 1.33f );          /* centre sphere */
	for ( i = 1; i <= N_ORBIT; ++i )
	{
		const float r = (float) ( demo_rand() % 16365 ) / 16384.0f, g = (float) ( demo_rand() % 16365 ) / 16384.0f,
			b = (float) ( demo_rand() % 16365 ) / 16384.0f;
		material( &sc->mats[1 + i], 1.0f, r, g, b, 0.05f, 0.5f, 15.0f, 0.5f, 0.5f, 0.0f, 1.0f );
		sc->koef[i] = (float) ( demo_rand() % 101 );
	}
	material( &sc->mats[11], 1.0f, 1.0f, 1.0f, 1.0f, 0.05f, 1.0f, 5.0f, 1.0f, 0.0f, 0.0f, 1.0f );           /* box */
	rt_vector3_create( &lp, -20.0f, -35.0f, 0.0f );
	rt_color_create( &lc, 1.0f, 1.0f, 1.0f, 1.0f );
	rt_point_light_create( &sc->light, &lp, 100.0f, &lc );
	build_box( sc );
	rt_mesh_reserve( &sc->water, (rt_ulong) ( squares + 1 ) * ( squares + 1 ), (rt_ulong) 2 * squares * squares, 1 );
}

/* one frame's worth of scene submission, in the reference's order (main.c:674-716) */
static void demo_submit( rt_render_pipe *rp, demo_scene *sc, float t, rt_mesh *extra )
{
	int i;
	rt_sphere s;
	rt_vector3 p;
	const double radius = 25.0 + 25.0 * ( 0.5 + sin( 10.0 * (double) t ) * 0.5 );
	for ( i = 0; i < 12; ++i )
		rt_render_pipe_add_material( rp, &sc->mats[i], (unsigned long) i );
	rt_render_pipe_add_light( rp, &sc->light, RT_LT_POINT );
	rt_vector3_create( &p, 0.0f, 0.0f, 50.0f );
	rt_sphere_create( &s, &p, 10.0f, 1 );
	rt_render_pipe_add_primitive( rp, &s, RT_PT_SPHERE );
	for ( i = 1; i <= N_ORBIT; ++i )
	{
		const double a = (double) t + ( i - 1 ) * 2.0 / 9.0 * M_PI, kf = sc->koef[i];
		rt_vector3_create( &p, (float) ( sin( a ) * radius ), (float) ( kf * 0.5 * sin( kf * 0.5 * (double) t ) ),
			(float) ( 50.0 + cos( a ) * radius ) );
		rt_sphere_create( &s, &p, 3.66541f, (unsigned long) ( 1 + i ) );
		rt_render_pipe_add_primitive( rp, &s, RT_PT_SPHERE );
	}
	rt_mesh_add( rp, &sc->box, 11 );
	build_water( sc, t );
	rt_mesh_add( rp, &sc->water, 0 );
	if ( extra )
		rt_mesh_add( rp, extra, 11 );
}

/* ---- image files ------------------------------------------------------- */

static int write_ppm( const char *path, const rt_argb *px, int w, int h )
{
	FILE *f = fopen( path, "wb" );
	long i;
	if ( !f ) return -1;
	fprintf( f, "P6\n%d %d\n255\n", w, h );
	for ( i = 0; i < (long) w * h; ++i )
	{
		unsigned char rgb[3] = { px[i].r, px[i].g, px[i].b };
		fwrite( rgb, 1, 3, f );
	}
	return fclose( f );
}

static void png_chunk( FILE *f, const char *tag, const unsigned char *data, unsigned int len )
{
	unsigned char b[4] = { (unsigned char) ( len >> 24 ), (unsigned char) ( len >> 16 ), (unsigned char) ( len >> 8 ), (unsigned char) len };
	unsigned long crc = crc32( 0L, (const unsigned char *) tag, 4 );
	fwrite( b, 1, 4, f );
	fwrite( tag, 1, 4, f );
	if ( len )
	{
		fwrite( data, 1, len, f );
		crc = crc32( crc, data, len );
	}
	b[0] = (unsigned char) ( crc >> 24 ); b[1] = (unsigned char) ( crc >> 16 ); b[2] = (unsigned char) ( crc >> 8 ); b[3] = (unsigned char) crc;
	fwrite( b, 1, 4, f );
}

/* 8-bit RGB PNG, one deflate stream, filter 0 on every row */
static int write_png( const char *path, const rt_argb *px, int w, int h )
{
	static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
	const size_t rawLen = (size_t) h * ( 1 + 3 * (size_t) w );
	unsigned char *raw = (unsigned char *) malloc( rawLen ), *z, hdr[13];
	uLongf zLen = compressBound( rawLen );
	FILE *f;
	int x, y;
	if ( !raw ) return -1;
	for ( y = 0; y < h; ++y )
	{
		unsigned char *row = raw + (size_t) y * ( 1 + 3 * (size_t) w );
		*row++ = 0;
		for ( x = 0; x < w; ++x )
		{
			const rt_argb p = px[(size_t) y * w + x];
			*row++ = p.r; *row++ = p.g; *row++ = p.b;
		}
	}
	z = (unsigned char *) malloc( zLen );
	if ( !z || compress2( z, &zLen, raw, rawLen, 1 ) != Z_OK || !( f = fopen( path, "wb" ) ) )
	{
		free( raw ); free( z );
		return -1;
	}
	fwrite( sig, 1, 8, f );
	hdr[0] = (unsigned char) ( w >> 24 ); hdr[1] = (unsigned char) ( w >> 16 ); hdr[2] = (unsigned char) ( w >> 8 ); hdr[3] = (unsigned char) w;
	hdr[4] = (unsigned char) ( h >> 24 ); hdr[5] = (unsigned char) ( h >> 16 ); hdr[6] = (unsigned char) ( h >> 8 ); hdr[7] = (unsigned char) h;
	hdr[8] = 8; hdr[9] = 2; hdr[10] = 0; hdr[11] = 0; hdr[12] = 0;
	png_chunk( f, "IHDR", hdr, 13 );
	png_chunk( f, "IDAT", z, (unsigned int) zLen );
	png_chunk( f, "IEND", NULL, 0 );
	free( raw ); free( z );
	return fclose( f );
}

/* ---- save file: float t, float dT, unsigned long next frame number -------- */

static int load_save( float *t, float *dT, unsigned long *n )
{
	FILE *f = fopen( "save", "rb" );
	int ok;
	if ( !f ) return 0;
	ok = fread( t, sizeof( float ), 1, f ) == 1 && fread( dT, sizeof( float ), 1, f ) == 1
		&& fread( n, sizeof( unsigned long ), 1, f ) == 1;
	fclose( f );
	return ok;
}

static void store_save( float t, float dT, unsigned long n )
{
	FILE *f = fopen( "save", "wb" );
	if ( !f ) return;
	fwrite( &t, sizeof( float ), 1, f );
	fwrite( &dT, sizeof( float ), 1, f );
	fwrite( &n, sizeof( unsigned long ), 1, f );
	fclose( f );
}

static double now( void )
{
	struct timespec ts;
	clock_gettime( CLOCK_MONOTONIC, &ts );
	return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

int main( int argc, char **argv )
{
	int frames = 1, w = 1280, h = 768, depth = 0, leaf = 0, stack = 0, squares = 150, device = 0, resume = 0, gpuBuild = -1, prune = -1, i;
	float t = 0.0f, dT = 0.0001f;
	unsigned long pngN = 0;
	const char *out = "pngs", *format = "png", *obj = NULL;
	rt_render_pipe rp;
	rt_camera cam;
	rt_matrix4 world;
	demo_scene sc;
	rt_mesh extra;
	double tDraw = 0.0;

	for ( i = 1; i < argc; ++i )
	{
		const char *a = argv[i], *v = i + 1 < argc ? argv[i + 1] : "";
		if ( !strcmp( a, "--frames" ) ) { frames = atoi( v ); ++i; }
		else if ( !strcmp( a, "--size" ) ) { if ( sscanf( v, "%dx%d", &w, &h ) != 2 ) { fprintf( stderr, "--size WxH\n" ); return 2; } ++i; }
		else if ( !strcmp( a, "--t" ) ) { t = (float) atof( v ); ++i; }
		else if ( !strcmp( a, "--dt" ) ) { dT = (float) atof( v ); ++i; }
		else if ( !strcmp( a, "--out" ) ) { out = v; ++i; }
		else if ( !strcmp( a, "--format" ) ) { format = v; ++i; }
		else if ( !strcmp( a, "--depth" ) ) { depth = atoi( v ); ++i; }
		else if ( !strcmp( a, "--leaf" ) ) { leaf = atoi( v ); ++i; }
		else if ( !strcmp( a, "--stack" ) ) { stack = atoi( v ); ++i; }
		else if ( !strcmp( a, "--squares" ) ) { squares = atoi( v ); ++i; }
		else if ( !strcmp( a, "--obj" ) ) { obj = v; ++i; }
		else if ( !strcmp( a, "--device" ) ) { device = atoi( v ); ++i; }
		else if ( !strcmp( a, "--resume" ) ) resume = 1;
		else if ( !strcmp( a, "--gpu-build" ) ) gpuBuild = 1;
		else if ( !strcmp( a, "--host-build" ) ) gpuBuild = 0;
		else if ( !strcmp( a, "--prune-leaves" ) ) prune = 1;                 /* (the pipe's default) */
		else if ( !strcmp( a, "--reference-leaves" ) ) prune = 0;
		else { fprintf( stderr, "unknown option %s\n", a ); return 2; }
	}
	if ( frames < 1 || w < 2 || h < 2 || squares < 1 ) { fprintf( stderr, "bad arguments\n" ); return 2; }
	if ( resume && load_save( &t, &dT, &pngN ) )
		t += dT;                                             /* main.c:793-800 */
	if ( strcmp( format, "none" ) )
		mkdir( out, 0777 );

	rt_init( argv[0] );
	rt_render_pipe_select_device( device );
	rt_render_pipe_create( &rp, w, h );
	if ( depth ) rt_render_pipe_set_param( &rp, RT_PARAM_MAX_DEPTH, depth );
	if ( leaf ) rt_render_pipe_set_param( &rp, RT_PARAM_MAX_PRIMS_IN_NODE, leaf );
	if ( stack ) rt_render_pipe_set_param( &rp, RT_PARAM_STACK_SIZE, stack );
	if ( gpuBuild >= 0 ) rt_render_pipe_set_param( &rp, RT_PARAM_GPU_BUILD, gpuBuild );
	if ( prune >= 0 ) rt_render_pipe_set_param( &rp, RT_PARAM_PRUNE_LEAVES, prune );

	/* camera of the demo (main.c:780,815-825) */
	rt_camera_create( &cam, (float) w / (float) h, 0.125f * (float) M_PI );
	memset( &world, 0, sizeof( world ) );
	world._11 = 1.0f;
	world._22 = 0.968582f; world._23 = -0.248690f;
	world._32 = 0.248690f; world._33 = 0.968582f;
	world._42 = -12.0f; world._43 = -43.0f; world._44 = 1.0f;
	rt_camera_set_view_matrix( &cam, &world );
	rt_render_pipe_set_camera( &rp, &cam );

	demo_init( &sc, squares );
	memset( &extra, 0, sizeof( extra ) );
	if ( obj && rt_mesh_load_from_obj( &extra, (char *) obj ) < 0 )
	{
		fprintf( stderr, "obj load error: %s\n", obj );
		obj = NULL;
	}

	for ( i = 0; i < frames; ++i )
	{
		char path[1024];
		double t0;
		rt_argb *px;
		demo_submit( &rp, &sc, t, obj ? &extra : NULL );
		t0 = now();
		px = rt_render_pipe_draw( &rp );
		tDraw += now() - t0;
		if ( !strcmp( format, "png" ) )
		{
			snprintf( path, sizeof( path ), "%s/%lu.png", out, pngN );
			if ( write_png( path, px, w, h ) ) { fprintf( stderr, "cannot write %s\n", path ); return 1; }
		}
		else if ( !strcmp( format, "ppm" ) )
		{
			snprintf( path, sizeof( path ), "%s/%lu.ppm", out, pngN );
			if ( write_ppm( path, px, w, h ) ) { fprintf( stderr, "cannot write %s\n", path ); return 1; }
		}
		++pngN;
		store_save( t, dT, pngN );
		if ( t > 2.0f * (float) M_PI )                       /* main.c:76-86 */
			t = 0.0f;
		t += dT;
	}
	printf( "%d frame(s) %dx%d, %lu triangles, %.2f ms per draw (tree build + upload + render + read-back)\n",
		frames, w, h, (unsigned long) ( sc.water.tc + sc.box.tc + ( obj ? extra.tc : 0 ) ), 1e3 * tDraw / frames );

	rt_mesh_release( &sc.box );
	rt_mesh_release( &sc.water );
	if ( obj ) rt_mesh_release( &extra );
	rt_render_pipe_cleanup( &rp );
	return 0;
}
