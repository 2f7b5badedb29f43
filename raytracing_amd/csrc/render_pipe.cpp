// Render-pipe API of include/rt_funcs_render_pipe.h on top of the HIP layer.
//
// Scene data added through rt_render_pipe_add_* is kept in host arrays (the
// reference wrote it to OpenCL buffers and read triangles and vertices back
// for every tree build, rt_funcs_render_pipe.c:1106-1112); rt_render_pipe_draw
// builds the kd-tree straight into the HBM layout, uploads what changed and
// launches the kernels.  With RT_PARAM_TREE_CACHE the tree and its device
// copy are reused while the geometry bytes are unchanged.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rt_funcs_render_pipe.h"
#include "rt_kdtree.h"

char *rt_cl_include_path = nullptr;
char *rt_cl_raytrace_kernel_path = nullptr;

// Growing array for the geometry handed to the tree builder. From 1 MiB up it
// lives in page-locked memory where the driver provides it: the GPU builder's
// upload of 1 M triangles then runs at the bus rate instead of through a
// staging copy. clear() keeps the block, so a frame loop allocates once.
template <class T> struct GeomBuf
{
	static constexpr size_t PIN_FROM = 1u << 20;
	rt_hip_ctx *hip = nullptr;
	T *p = nullptr;
	size_t n = 0, cap = 0;
	bool pinned = false;

	T *data() { return p; }
	size_t size() const { return n; }
	void clear() { n = 0; }
	void reserve( size_t want )
	{
		if ( want <= cap )
			return;
		size_t c = cap ? cap * 2 : 64;
		if ( c < want ) c = want;
		T *q = nullptr;
		bool qPinned = false;
		if ( hip && c * sizeof( T ) >= PIN_FROM )
		{
			q = (T *) rt_hip_alloc_host( hip, c * sizeof( T ) );
			qPinned = q != nullptr;
		}
		if ( !q )
			q = (T *) malloc( c * sizeof( T ) );
		if ( !q )
		{
			fprintf( stderr, "rt: out of memory (%zu bytes of geometry)\n", c * sizeof( T ) );
			exit( -3 );
		}
		if ( n )
			memcpy( q, p, n * sizeof( T ) );
		drop();
		p = q;
		cap = c;
		pinned = qPinned;
	}
	void append( const T *src, size_t k )
	{
		if ( !k )
			return;
		reserve( n + k );
		memcpy( p + n, src, k * sizeof( T ) );
		n += k;
	}
	T *grow( size_t k )      // k uninitialised entries at the end
	{
		reserve( n + k );
		T *at = p + n;
		n += k;
		return at;
	}
	void drop()
	{
		if ( pinned )
			rt_hip_free_host( hip, p );
		else
			free( p );
		p = nullptr;
		cap = 0;
	}
	void release() { drop(); n = 0; }
};

// Change detector of the tree cache: FNV-1a steps over 8-byte words in eight independent lanes (one multiply chain per
// lane: the latency of the multiply, not its throughput, is what a single chain would pay), folded into 64 bits.
struct GeomHash
{
	uint64_t lane[8];
	uint64_t words;
	GeomHash() { reset(); }
	void reset()
	{
		for ( int k = 0; k < 8; ++k )
			lane[k] = 1469598103934665603ull + (uint64_t) k;
		words = 0;
	}
	static inline uint64_t step( uint64_t h, uint64_t w ) { return ( h ^ w ) * 1099511628211ull; }
	// n words; the lane of a word follows from its place in the whole sequence, so the split into calls does not matter
	void add( const uint64_t *w, size_t n )
	{
		size_t i = 0;
		for ( ; i < n && ( ( words + i ) & 7u ) != 0u; ++i )
			lane[( words + i ) & 7u] = step( lane[( words + i ) & 7u], w[i] );
		for ( ; i + 8 <= n; i += 8 )
			for ( int k = 0; k < 8; ++k )
				lane[k] = step( lane[k], w[i + k] );
		for ( ; i < n; ++i )
			lane[( words + i ) & 7u] = step( lane[( words + i ) & 7u], w[i] );
		words += n;
	}
	void align() { words = ( words + 7u ) & ~(uint64_t) 7u; }      // the next word goes to lane 0
	uint64_t value() const
	{
		uint64_t h = step( 1469598103934665603ull, words );
		for ( int k = 0; k < 8; ++k )
			h = step( h, lane[k] );
		return h;
	}
};
static_assert( sizeof( rt_vertex ) % 8 == 0 && sizeof( rt_triangle ) == 32, "geometry is hashed in 8-byte words" );

struct rt_pipe_impl
{
	rt_hip_ctx *hip = nullptr;
	GeomBuf<rt_vertex> verts;
	GeomBuf<rt_triangle> tris;
	std::vector<rt_sphere> spheres;
	std::vector<rt_point_light> lights;
	std::vector<rt_material> mats;
	rt_camera cam;
	int maxDepth = MAX_DEPTH, maxPrims = MAX_PRIMS_IN_NODE, stackSize = STACK_SIZE;
	// pruneLeaves: leaves list only the triangles that reach their cell (same frame, fewer triangle tests; the
	// reference's complete lists are kept for the rays whose walk depends on them) - on unless switched off
	int buildThreads = 0, treeCache = 0, pruneLeaves = 1;
	int gpuBuild = 2;                 // 0 = host builder, 1 = GPU builder, 2 = GPU builder from GPU_BUILD_MIN triangles up
	int bandH = 16, nParts = 1, part = 0;
	bool matsDirty = true;
	bool haveTree = false;
	bool screenPinned = false;
	uint64_t geomHash = 0, geomBytes = 0;
	// change detector of the tree cache, kept up to date while the geometry is added (the words are in registers or
	// in cache then: a pass of its own over 44 MB of unchanged geometry took 6 ms, twice the GPU build it saves)
	GeomHash adding;
	bool addingValid = true;         // every add_triangles call since the last draw went into `adding`
	// Geometry goes to the device while it is being added, when the draw is going to build its tree there and is not
	// going to reuse the last one (a 1 M-triangle scene is 44 MB: 0.8 ms on the bus that the draw would wait for):
	// how many vertices / triangles have been handed to rt_hip_stage_geometry since the last draw
	size_t stagedV = 0, stagedT = 0;
	bool staging = false, stagingLost = false;
	rt_kdtree_flat lastInfo;          // counts of the tree on the device (arrays released)
};

static int g_device = 0;
static const size_t GPU_BUILD_MIN = 8192;
static const size_t STAGE_BYTES = 4u << 20;      // geometry goes to the device in pieces of this size while it is added
static const size_t STAGE_TRIS = STAGE_BYTES / sizeof( rt_triangle );

static void die_hip( const char *what )
{
	fprintf( stderr, "rt: %s: %s\n", what, rt_hip_last_error() );
	fflush( stderr );
	exit( -3 );
}

// Hands what has been added since the last call to the device, once it is a piece worth a copy of its own.
static void stage_added( rt_pipe_impl *im )
{
	if ( im->gpuBuild == 0 || im->treeCache || im->stagingLost || im->tris.size() < GPU_BUILD_MIN )
		return;
	const size_t fresh = ( im->verts.size() - im->stagedV ) * sizeof( rt_vertex ) + ( im->tris.size() - im->stagedT ) * sizeof( rt_triangle );
	if ( fresh < STAGE_BYTES )
		return;
	const int rc = rt_hip_stage_geometry( im->hip, im->verts.data(), im->verts.size(), im->tris.data(), im->tris.size() );
	if ( rc == 1 )
		die_hip( "geometry upload" );
	if ( rc == 2 )
		im->stagingLost = true;          // (another pipe of this thread uses the builder: this frame's build copies everything)
	im->stagedV = im->verts.size();
	im->stagedT = im->tris.size();
	im->staging = true;
}

// Copies on their way read the geometry arrays: before the arrays move or go, they have to be done.
static void staged_copies_done( rt_pipe_impl *im )
{
	if ( im->staging && rt_hip_sync( im->hip ) )
		die_hip( "geometry upload" );
}

static rt_pipe_impl *impl_of( rt_render_pipe *pRp )
{
	if ( !pRp || !pRp->impl )
		exit( -1 );
	return pRp->impl;
}

extern "C" {

// derives the executable's directory, as the reference's rt_init (:3-19) does
// to find its kernel sources; nothing is loaded from it here
void rt_init( char *path )
{
	if ( !path )
		exit( -1 );
	size_t cut = strlen( path );
	while ( cut > 0 && path[cut - 1] != '/' )
		--cut;
	if ( cut > 0 )
		--cut;                       // drop the slash itself
	static const char tail[] = "/CL/rt_raytrace.cl";
	free( rt_cl_include_path );
	free( rt_cl_raytrace_kernel_path );
	rt_cl_include_path = (char *) malloc( cut + 1 );
	rt_cl_raytrace_kernel_path = (char *) malloc( cut + sizeof( tail ) );
	memcpy( rt_cl_include_path, path, cut );
	rt_cl_include_path[cut] = '\0';
	memcpy( rt_cl_raytrace_kernel_path, path, cut );
	memcpy( rt_cl_raytrace_kernel_path + cut, tail, sizeof( tail ) );
}

void rt_render_pipe_select_device( int device ) { g_device = device; }

void rt_init_buffers( rt_render_pipe *pRp )
{
	rt_pipe_impl *im = impl_of( pRp );
	pRp->primBlocks = pRp->primDecsBlocks = pRp->lightBlocks = pRp->lightDecsBlocks = 1;
	pRp->triangleBlocks = pRp->vertexBlocks = pRp->materialBlocks = 1;
	pRp->primsEnd = pRp->lightsEnd = 0;
	pRp->primsCount = pRp->lightsCount = pRp->trianglesCount = pRp->vertexCount = 0;
	if ( im->staging )
	{
		staged_copies_done( im );
		rt_hip_unstage_geometry( im->hip );
		im->staging = false;
	}
	im->stagingLost = false;
	im->stagedV = im->stagedT = 0;
	im->verts.clear();
	im->tris.clear();
	im->adding.reset();
	im->addingValid = true;
	im->spheres.clear();
	im->lights.clear();
	im->mats.assign( MATERIALS_IN_BLOCK, rt_material() );
	memset( im->mats.data(), 0, sizeof( rt_material ) * im->mats.size() );
	im->matsDirty = true;
}

void rt_render_pipe_create( rt_render_pipe *pRp, int w, int h )
{
	if ( !pRp )
		exit( -1 );
	memset( pRp, 0, sizeof( *pRp ) );
	pRp->impl = new rt_pipe_impl();
	memset( &pRp->impl->lastInfo, 0, sizeof( rt_kdtree_flat ) );
	pRp->impl->hip = rt_hip_create( g_device );
	if ( !pRp->impl->hip )
		die_hip( "cannot create the HIP context" );
	pRp->impl->verts.hip = pRp->impl->tris.hip = pRp->impl->hip;
	if ( rt_hip_set_param( pRp->impl->hip, RT_PARAM_PRUNE_LEAVES, pRp->impl->pruneLeaves ) )
		die_hip( "rt_render_pipe_create" );
	rt_init_buffers( pRp );
	{
		rt_camera &c = pRp->impl->cam;
		memset( &c, 0, sizeof( c ) );
		c.world._11 = c.world._22 = c.world._33 = c.world._44 = 1.0f;
		c.viewToPersp = c.world;
	}
	pRp->screenData = NULL;
	rt_render_pipe_set_image_size( pRp, w, h );
}

static void release_screen( rt_render_pipe *pRp, rt_pipe_impl *im )
{
	if ( im->screenPinned )
		rt_hip_free_host( im->hip, pRp->screenData );
	else
		free( pRp->screenData );
	im->screenPinned = false;
	pRp->screenData = NULL;
}

void rt_render_pipe_set_image_size( rt_render_pipe *pRp, int w, int h )
{
	rt_pipe_impl *im = impl_of( pRp );
	if ( rt_hip_set_image( im->hip, w, h ) )
		die_hip( "rt_render_pipe_set_image_size" );
	release_screen( pRp, im );
	const size_t bytes = sizeof( rt_argb ) * (size_t) w * (size_t) h;
	// page-locked where the driver allows it: the read-back then needs no staging copy
	pRp->screenData = (rt_argb *) rt_hip_alloc_host( im->hip, bytes );
	im->screenPinned = pRp->screenData != NULL;
	if ( !pRp->screenData )
		pRp->screenData = (rt_argb *) malloc( bytes );
	pRp->w = w;
	pRp->h = h;
}

void rt_render_pipe_reset_blocks( rt_render_pipe *pRp )
{
	impl_of( pRp );
	rt_init_buffers( pRp );
}

void rt_render_pipe_set_camera( rt_render_pipe *pRp, rt_camera *pFr )
{
	rt_pipe_impl *im = impl_of( pRp );
	if ( !pFr )
		exit( -1 );
	im->cam = *pFr;
}

void rt_render_pipe_get_camera( rt_render_pipe *pRp, rt_camera **ppC )
{
	rt_pipe_impl *im = impl_of( pRp );
	if ( !ppC )
		exit( -1 );
	*ppC = &im->cam;
}

void rt_render_pipe_free_camera( rt_render_pipe *pRp, rt_camera *pC )
{
	impl_of( pRp );
	if ( !pC )
		exit( -1 );
	// the camera is read at the next draw; nothing to write back
}

// block accounting of rt_add_block (:186-207): grow by whole blocks
static void grow_blocks( rt_ulong *blocks, size_t blockSize, size_t needed )
{
	rt_ulong add = needed / blockSize;
	*blocks += add ? add : 1;
}

void rt_render_pipe_add_primitive( rt_render_pipe *pRp, void *pPrim, RT_PRIMITIVE_TYPE pt )
{
	rt_pipe_impl *im = impl_of( pRp );
	if ( !pPrim )
		exit( -1 );
	if ( ( pRp->primsCount + 1 ) > PRIMS_IN_BLOCK * pRp->primDecsBlocks )
		grow_blocks( &pRp->primDecsBlocks, sizeof( rt_prim_desc ) * PRIMS_IN_BLOCK, sizeof( rt_prim_desc ) );
	if ( pt == RT_PT_SPHERE )
	{
		if ( ( pRp->primsEnd + sizeof( rt_sphere ) ) >= (size_t) PRIMS_BLOCK_SIZE * pRp->primBlocks )
			grow_blocks( &pRp->primBlocks, PRIMS_BLOCK_SIZE, sizeof( rt_sphere ) );
		im->spheres.push_back( *(const rt_sphere *) pPrim );
		pRp->primsEnd += sizeof( rt_sphere );
	}
	// Other primitive types: the reference records a descriptor and counts the
	// primitive (:228-229, :258) but never uploads or intersects it (:234-256),
	// so only the counter moves here.
	++pRp->primsCount;
}

void rt_render_pipe_add_triangles( rt_render_pipe *pRp, rt_vertex *pV, rt_triangle *pTr, rt_ulong vOffset,
	rt_ulong tOffset, rt_ulong vCount, rt_ulong tCount, rt_ulong tGroup )
{
	rt_pipe_impl *im = impl_of( pRp );
	if ( !pV || !pTr )
		exit( -1 );
	if ( ( pRp->trianglesCount + tCount ) >= (rt_ulong) TRIANGLES_IN_BLOCK * pRp->triangleBlocks )
		grow_blocks( &pRp->triangleBlocks, TRIANGLES_IN_BLOCK * sizeof( rt_triangle ), sizeof( rt_triangle ) * tCount );
	if ( ( pRp->vertexCount + vCount ) >= (rt_ulong) VERTEX_IN_BLOCK * pRp->vertexBlocks )
		grow_blocks( &pRp->vertexBlocks, VERTEX_IN_BLOCK * sizeof( rt_vertex ), sizeof( rt_vertex ) * vCount );

	const rt_ulong base = pRp->vertexCount;
	if ( im->verts.size() + vCount > im->verts.cap || im->tris.size() + tCount > im->tris.cap )
		staged_copies_done( im );          // (the arrays are about to move)
	im->verts.append( pV + vOffset, vCount );
	const bool hashed = im->treeCache && im->addingValid;
	if ( hashed )
	{
		// every word written also goes into the change detector (vertices first, as they lie in memory)
		im->adding.add( (const uint64_t *) ( pV + vOffset ), vCount * ( sizeof( rt_vertex ) / 8 ) );
		im->adding.align();
	}
	else
		im->addingValid = false;
	stage_added( im );
	im->tris.reserve( im->tris.size() + tCount );
	// (in pieces, so that the first of a large mesh is on the bus while the next is being copied)
	for ( rt_ulong done = 0; done < tCount; )
	{
		const rt_ulong n = tCount - done < (rt_ulong) STAGE_TRIS ? tCount - done : (rt_ulong) STAGE_TRIS;
		const rt_triangle *src = pTr + tOffset + done;
		rt_triangle *dst = im->tris.grow( n );
		rt_ulong i = 0;
		if ( hashed )
		{
			// (a piece holds an even number of triangles unless it is the last: 8 words per pair keep the lanes in step)
			GeomHash &H = im->adding;
			for ( ; i + 2 <= n; i += 2 )
				for ( int u = 0; u < 2; ++u )
				{
					const rt_triangle &s = src[i + u];
					rt_triangle &d = dst[i + u];
					d.pV0 = base + s.pV0 - vOffset;
					d.pV1 = base + s.pV1 - vOffset;
					d.pV2 = base + s.pV2 - vOffset;
					d.mat = tGroup;
					H.lane[4 * u] = GeomHash::step( H.lane[4 * u], d.pV0 );
					H.lane[4 * u + 1] = GeomHash::step( H.lane[4 * u + 1], d.pV1 );
					H.lane[4 * u + 2] = GeomHash::step( H.lane[4 * u + 2], d.pV2 );
					H.lane[4 * u + 3] = GeomHash::step( H.lane[4 * u + 3], d.mat );
				}
			H.words += 4 * i;
		}
		for ( ; i < n; ++i )
		{
			const rt_triangle &s = src[i];
			dst[i].pV0 = base + s.pV0 - vOffset;
			dst[i].pV1 = base + s.pV1 - vOffset;
			dst[i].pV2 = base + s.pV2 - vOffset;
			dst[i].mat = tGroup;
			if ( hashed )
				im->adding.add( (const uint64_t *) &dst[i], 4 );
		}
		done += n;
		stage_added( im );
	}
	pRp->trianglesCount += tCount;
	pRp->vertexCount += vCount;
}

void rt_render_pipe_add_material( rt_render_pipe *pRp, rt_material *pMat, unsigned long int ind )
{
	rt_pipe_impl *im = impl_of( pRp );
	if ( !pMat )
		exit( -1 );
	while ( ind >= (unsigned long) MATERIALS_IN_BLOCK * pRp->materialBlocks )
		grow_blocks( &pRp->materialBlocks, MATERIALS_IN_BLOCK * sizeof( rt_material ), sizeof( rt_material ) );
	if ( im->mats.size() < (size_t) MATERIALS_IN_BLOCK * pRp->materialBlocks )
	{
		rt_material z;
		memset( &z, 0, sizeof( z ) );
		im->mats.resize( (size_t) MATERIALS_IN_BLOCK * pRp->materialBlocks, z );
	}
	if ( memcmp( &im->mats[ind], pMat, sizeof( rt_material ) ) != 0 )
	{
		im->mats[ind] = *pMat;
		im->matsDirty = true;
	}
}

void rt_render_pipe_add_light( rt_render_pipe *pRp, void *pLight, RT_LIGHT_TYPE lt )
{
	rt_pipe_impl *im = impl_of( pRp );
	if ( !pLight )
		exit( -1 );
	if ( ( pRp->lightsCount + 1 ) > LIGHTS_IN_BLOCK * pRp->lightDecsBlocks )
		grow_blocks( &pRp->lightDecsBlocks, sizeof( rt_light_desc ) * LIGHTS_IN_BLOCK, sizeof( rt_light_desc ) );
	if ( lt == RT_LT_POINT )
	{
		if ( ( pRp->lightsEnd + sizeof( rt_point_light ) ) >= (size_t) LIGHTS_BLOCK_SIZE * pRp->lightBlocks )
			grow_blocks( &pRp->lightBlocks, LIGHTS_BLOCK_SIZE, sizeof( rt_point_light ) );
		im->lights.push_back( *(const rt_point_light *) pLight );
		pRp->lightsEnd += sizeof( rt_point_light );
		++pRp->lightsCount;
	}
	// directed lights are declared by the reference but never shaded (:460)
}

void rt_render_pipe_set_param( rt_render_pipe *pRp, int key, int value )
{
	rt_pipe_impl *im = impl_of( pRp );
	switch ( key )
	{
	case RT_PARAM_STACK_SIZE:
		if ( rt_hip_set_param( im->hip, key, value ) )
			die_hip( "rt_render_pipe_set_param" );
		im->stackSize = value;
		break;
	case RT_PARAM_MAX_DEPTH:
		if ( value < 0 || value > RT_HIP_MAX_TREE_DEPTH ) { fprintf( stderr, "rt: RT_PARAM_MAX_DEPTH out of range\n" ); exit( -3 ); }
		im->maxDepth = value;
		im->haveTree = false;
		break;
	case RT_PARAM_MAX_PRIMS_IN_NODE:
		if ( value < 1 ) { fprintf( stderr, "rt: RT_PARAM_MAX_PRIMS_IN_NODE out of range\n" ); exit( -3 ); }
		im->maxPrims = value;
		im->haveTree = false;
		break;
	case RT_PARAM_BUILD_THREADS: im->buildThreads = value; break;
	case RT_PARAM_TREE_CACHE: im->treeCache = value; break;
	case RT_PARAM_GPU_BUILD:
		if ( value < 0 || value > 2 ) { fprintf( stderr, "rt: RT_PARAM_GPU_BUILD must be 0, 1 or 2\n" ); exit( -3 ); }
		im->gpuBuild = value;
		im->haveTree = false;
		break;
	case RT_PARAM_PRUNE_LEAVES:
		if ( rt_hip_set_param( im->hip, key, value ) )
			die_hip( "rt_render_pipe_set_param" );
		im->pruneLeaves = value ? 1 : 0;
		im->haveTree = false;
		break;
	case RT_PARAM_OVERLAP:
	case RT_PARAM_COOP_LIVE:
	case RT_PARAM_COOP_STACK:
	case RT_PARAM_LAUNCH_WAVES:
		if ( rt_hip_set_param( im->hip, key, value ) )
			die_hip( "rt_render_pipe_set_param" );
		break;
	default:
		fprintf( stderr, "rt: unknown parameter %d\n", key );
		exit( -3 );
	}
}

void rt_render_pipe_set_tile( rt_render_pipe *pRp, int bandH, int nParts, int part )
{
	rt_pipe_impl *im = impl_of( pRp );
	if ( rt_hip_set_tile( im->hip, bandH, nParts, part ) )
		die_hip( "rt_render_pipe_set_tile" );
	im->bandH = bandH; im->nParts = nParts; im->part = part;
}

rt_hip_ctx *rt_render_pipe_hip( rt_render_pipe *pRp ) { return impl_of( pRp )->hip; }

void rt_kdtree_build( rt_render_pipe *pRp, rt_box *pBoundingBox, rt_cl_kdtree_node **nodes, rt_ulong *nodesCount,
	rt_ulong **primsIdx, rt_ulong *primsIdxCount )
{
	rt_pipe_impl *im = impl_of( pRp );
	if ( !pBoundingBox || !nodes || !nodesCount || !primsIdx || !primsIdxCount )
		exit( -1 );
	rt_kdtree_flat t;
	int rc = rt_kdtree_build_flat( im->verts.data(), im->tris.data(), im->verts.size(), im->tris.size(),
		im->maxDepth, im->maxPrims, im->buildThreads, &t );
	if ( rc )
	{
		fprintf( stderr, "rt: kd-tree build failed (%d)\n", rc );
		exit( -3 );
	}
	*pBoundingBox = t.bbox;
	uint64_t nn = 0, ni = 0;
	rt_kdtree_flat_expand( &t, nodes, &nn, primsIdx, &ni );
	*nodesCount = nn;
	*primsIdxCount = ni;
	rt_kdtree_flat_release( &t );
}

// upload everything the kernels read and enqueue one frame
static void submit_frame( rt_render_pipe *pRp, void *d_out )
{
	rt_pipe_impl *im = impl_of( pRp );

	// the change detector is only paid for when a tree may be reused
	uint64_t h = 0;
	if ( im->treeCache )
	{
		if ( !im->addingValid )
		{
			// (the cache was switched on with geometry already added: one pass over it, in the order add_triangles takes)
			im->adding.reset();
			im->adding.add( (const uint64_t *) im->verts.data(), im->verts.size() * ( sizeof( rt_vertex ) / 8 ) );
			im->adding.align();
			im->adding.add( (const uint64_t *) im->tris.data(), im->tris.size() * ( sizeof( rt_triangle ) / 8 ) );
		}
		h = im->adding.value();
	}
	const uint64_t bytes = im->verts.size() * sizeof( rt_vertex ) + im->tris.size() * sizeof( rt_triangle );
	const bool reuse = im->treeCache && im->haveTree && h == im->geomHash && bytes == im->geomBytes;
	// small scenes are built faster by the host (the GPU builder costs about
	// 0.1 ms per tree level whatever the size); both give the same tree
	const bool onGpu = im->gpuBuild == 1 || ( im->gpuBuild == 2 && im->tris.size() >= GPU_BUILD_MIN );
	if ( im->staging && ( reuse || !onGpu ) )
	{
		// (a parameter changed between add and draw: what went ahead is not going to be used)
		staged_copies_done( im );
		rt_hip_unstage_geometry( im->hip );
	}
	if ( !reuse && onGpu )
	{
		// same tree, built in HBM: nothing but the raw geometry crosses PCIe
		if ( rt_hip_build_tree( im->hip, im->verts.data(), im->tris.data(), im->verts.size(), im->tris.size(),
			im->maxDepth, im->maxPrims ) )
			die_hip( "kd-tree build on the GPU" );
		im->haveTree = true;
		im->geomHash = h;
		im->geomBytes = bytes;
	}
	else if ( !reuse )
	{
		rt_kdtree_flat t;
		rt_kdtree_build_opts o;
		memset( &o, 0, sizeof( o ) );
		o.maxDepth = im->maxDepth;
		o.maxPrims = im->maxPrims;
		o.threads = im->buildThreads;
		o.pruneLeaves = im->pruneLeaves ? RT_PRUNE_LEAVES_MARGIN : 0.0f;
		int rc = rt_kdtree_build_flat_ex( im->verts.data(), im->tris.data(), im->verts.size(), im->tris.size(), &o, &t );
		if ( rc )
		{
			fprintf( stderr, "rt: kd-tree build failed (%d)\n", rc );
			exit( -3 );
		}
		if ( rt_hip_upload_tree_lists( im->hip, t.nodes, t.nNodes, t.recs, t.nRecs, t.nRecsFull, t.leafFull, t.shade, t.nTris, &t.bbox, t.depth ) )
			die_hip( "tree upload" );
		im->lastInfo = t;
		im->lastInfo.nodes = nullptr; im->lastInfo.recs = nullptr; im->lastInfo.shade = nullptr; im->lastInfo.leafFull = nullptr;
		rt_kdtree_flat_release( &t );
		im->haveTree = true;
		im->geomHash = h;
		im->geomBytes = bytes;
	}
	if ( im->matsDirty )
	{
		if ( rt_hip_upload_materials( im->hip, im->mats.data(), im->mats.size() ) )
			die_hip( "material upload" );
		im->matsDirty = false;
	}
	if ( rt_hip_upload_spheres( im->hip, im->spheres.data(), im->spheres.size() ) ) die_hip( "sphere upload" );
	if ( rt_hip_upload_lights( im->hip, im->lights.data(), im->lights.size() ) ) die_hip( "light upload" );
	if ( rt_hip_set_camera( im->hip, &im->cam ) ) die_hip( "camera" );
	if ( rt_hip_render( im->hip, d_out, 0 ) ) die_hip( "render" );
	// (the build has waited for its input)
	im->staging = im->stagingLost = false;
	im->stagedV = im->stagedT = 0;
}

// per-frame reset of :1226-1231: geometry, spheres and lights are consumed
static void consume_scene( rt_render_pipe *pRp )
{
	rt_pipe_impl *im = impl_of( pRp );
	im->verts.clear();
	im->tris.clear();
	im->spheres.clear();
	im->lights.clear();
	im->adding.reset();
	im->addingValid = true;
	pRp->trianglesCount = pRp->vertexCount = 0;
	pRp->lightsCount = pRp->primsCount = 0;
	pRp->primsEnd = pRp->lightsEnd = 0;
}

rt_argb *rt_render_pipe_draw( rt_render_pipe *pRp )
{
	rt_pipe_impl *im = impl_of( pRp );
	// rows rendered by this pipe land at the top of screenData (all of it
	// unless a tile split is set)
	if ( im->screenPinned )
	{
		// page-locked memory is mapped into the device's address space: the last kernel of the frame stores the
		// pixels straight into it across PCIe, which takes half as long as a copy of the finished frame behind it
		submit_frame( pRp, pRp->screenData );
		if ( rt_hip_sync( im->hip ) )
			die_hip( "frame" );
	}
	else
	{
		submit_frame( pRp, nullptr );
		if ( rt_hip_read_frame( im->hip, pRp->screenData ) )
			die_hip( "frame read-back" );
	}
	consume_scene( pRp );
	return pRp->screenData;
}

void rt_render_pipe_draw_device( rt_render_pipe *pRp, void *d_out )
{
	if ( !d_out )
		exit( -1 );
	submit_frame( pRp, d_out );
	consume_scene( pRp );
}

void rt_render_pipe_cleanup( rt_render_pipe *pRp )
{
	if ( !pRp )
		exit( -1 );
	if ( pRp->impl )
		release_screen( pRp, pRp->impl );
	else
		free( pRp->screenData );
	pRp->screenData = NULL;
	if ( pRp->impl )
	{
		staged_copies_done( pRp->impl );
		pRp->impl->verts.release();
		pRp->impl->tris.release();
		rt_hip_destroy( pRp->impl->hip );
		delete pRp->impl;
		pRp->impl = nullptr;
	}
}

} // extern "C"
