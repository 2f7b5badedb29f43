// kd-tree builder on the GPU (gfx950): the same tree as the host builder
// (kd_build.cpp, i.e. the reference's rule, rt_funcs_render_pipe.c:591-1209),
// produced level by level in HBM and emitted directly in the layout the
// traversal kernels read, so a rebuilt tree never crosses PCIe.
//
// One level = the set of nodes of one depth.  Per level:
//   k_decide    node  : leaf (count <= maxPrims or depth == maxDepth) or split axis + candidate planes
//   k_hist      ref   : histogram of every triangle extent against the 32 planes of its node
//   k_split     node  : counts left / right of every plane, SAH cost, lowest-index minimum
//   scan        node  : child node numbers, child reference ranges, leaf pool ranges (one hipCUB scan over triples)
//   k_children  node  : child boxes, next level's node table, this node's record of the tree
//   scan of flags + k_scatter   ref : STABLE partition of the triangle references
//                       (a reference goes left if min < sep, right if max > sep, possibly both)
// then, over the finished tree: sub-tree sizes bottom-up, pre-order positions
// top-down (left child = own index + 1), nodes + leaf records + shading records.
//
// fp32 expressions are the host builder's, operand for operand, and the file
// is compiled with -ffp-contract=off: the result is the host tree bit for bit
// (tests/test_kdtree_gpu.py).  The host drives the level loop and reads three
// totals per level to size the next one.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>

#include "rt_kdtree.h"

namespace {

constexpr int PARTS = SAH_PARTS;
constexpr int TB = 256;

#define GCHK( x ) do { hipError_t e_ = ( x ); if ( e_ != hipSuccess ) { \
	snprintf( g_gpu_build_err, sizeof( g_gpu_build_err ), "%s failed: %s (%s:%d)", #x, hipGetErrorString( e_ ), __FILE__, __LINE__ ); \
	return 10; } } while ( 0 )

thread_local char g_gpu_build_err[512] = "";

// a launch that is refused (grid, block or LDS size) says so at once, by name
#define LAUNCH_OR( cleanup, kernel, ... ) do { hipLaunchKernelGGL( kernel, __VA_ARGS__ ); hipError_t e_ = hipGetLastError(); if ( e_ != hipSuccess ) { \
	snprintf( g_gpu_build_err, sizeof( g_gpu_build_err ), "launch of %s failed: %s (%s:%d)", #kernel, hipGetErrorString( e_ ), __FILE__, __LINE__ ); \
	cleanup; return 10; } } while ( 0 )
#define LAUNCH( ... ) LAUNCH_OR( (void) 0, __VA_ARGS__ )

// device buffer that grows (contents are NOT kept unless keep > 0)
struct DVec
{
	void *p = nullptr;
	size_t cap = 0;
	int reserve( size_t bytes, hipStream_t s, size_t keep = 0 )
	{
		if ( bytes <= cap )
			return 0;
		size_t want = bytes + bytes / 2 + 4096;
		void *q = nullptr;
		GCHK( hipMalloc( &q, want ) );
		if ( keep && p )
			GCHK( hipMemcpyAsync( q, p, keep, hipMemcpyDeviceToDevice, s ) );
		if ( p )
		{
			GCHK( hipStreamSynchronize( s ) );
			GCHK( hipFree( p ) );
		}
		p = q;
		cap = want;
		return 0;
	}
	void release() { if ( p ) (void) hipFree( p ); p = nullptr; cap = 0; }
	template <class T> T *as() const { return (T *) p; }
};

// ---- per-triangle extents: the compare chain of rt_funcs_render_pipe.c:620-633
__device__ inline void tri_extent( float a, float b, float c, float &mn, float &mx )
{
	mx = a;
	if ( mx < b ) mx = b;
	if ( mx < c ) mx = c;
	mn = a;
	if ( mn > b ) mn = b;
	if ( mn > c ) mn = c;
}

// what the host learns about the geometry before the first level
struct GeomInfo
{
	uint32_t badIndex;     // some triangle names a vertex that does not exist
	uint32_t maxMat;       // highest material slot (saturated to 32 bits)
	float box[6];          // scene box: centre xyz, half extent xyz
	uint32_t lostInPlane;  // references that went to neither child of a split (triangles lying in the split plane)
};

// Extents of every triangle; the index check and the highest material slot ride
// along (the triangles are read here anyway). A triangle with a bad index gets
// no extents and raises the flag: the host gives up before anything else reads it.
__global__ __launch_bounds__( TB ) void k_extents( const rt_vertex *verts, const rt_triangle *tris, uint32_t nT, uint64_t nV, float *ext,
	GeomInfo *info )
{
	const uint32_t i = blockIdx.x * TB + threadIdx.x;
	uint32_t m = 0;
	if ( i < nT )
	{
		const rt_triangle t = tris[i];
		m = t.mat > 0xffffffffull ? 0xffffffffu : (uint32_t) t.mat;
		if ( t.pV0 >= nV || t.pV1 >= nV || t.pV2 >= nV )
			info->badIndex = 1u;
		else
		{
			const rt_vector3 p0 = verts[t.pV0].pos, p1 = verts[t.pV1].pos, p2 = verts[t.pV2].pos;
			tri_extent( p0.x, p1.x, p2.x, ext[i], ext[3ull * nT + i] );
			tri_extent( p0.y, p1.y, p2.y, ext[(size_t) nT + i], ext[4ull * nT + i] );
			tri_extent( p0.z, p1.z, p2.z, ext[2ull * nT + i], ext[5ull * nT + i] );
		}
	}
	for ( int off = 32; off; off >>= 1 )
	{
		const uint32_t o = (uint32_t) __shfl_xor( (int) m, off );
		m = m > o ? m : o;
	}
	if ( ( threadIdx.x & 63 ) == 0 && m )
		atomicMax( &info->maxMat, m );
}

// ---- scene box, rt_funcs_render_pipe.c:1114-1137 -------------------------
// The reference walks the vertices with "mn > p ? p : mn" starting from vertex 0:
// a NaN coordinate never replaces a number, and a NaN in vertex 0 stays for good.
// Minimum and maximum do not depend on the order otherwise (the two zeros differ
// in sign only, and the margin of 1 added afterwards hides that).
constexpr int BOX_BLOCKS = 1024;

__global__ __launch_bounds__( TB ) void k_box_partial( const rt_vertex *verts, uint64_t nV, float *part )
{
	const float inf = __builtin_huge_valf();
	float mn[3] = { inf, inf, inf }, mx[3] = { -inf, -inf, -inf };
	for ( uint64_t i = (uint64_t) blockIdx.x * TB + threadIdx.x; i < nV; i += (uint64_t) gridDim.x * TB )
	{
		const rt_vector3 p = verts[i].pos;
		const float c[3] = { p.x, p.y, p.z };
		for ( int a = 0; a < 3; ++a )
		{
			if ( c[a] < mn[a] ) mn[a] = c[a];
			if ( c[a] > mx[a] ) mx[a] = c[a];
		}
	}
	__shared__ float red[6][TB / 64];
	for ( int a = 0; a < 3; ++a )
	{
		for ( int off = 32; off; off >>= 1 )
		{
			const float o = __shfl_xor( mn[a], off ), q = __shfl_xor( mx[a], off );
			if ( o < mn[a] ) mn[a] = o;
			if ( q > mx[a] ) mx[a] = q;
		}
		if ( ( threadIdx.x & 63 ) == 0 )
		{
			red[a][threadIdx.x >> 6] = mn[a];
			red[3 + a][threadIdx.x >> 6] = mx[a];
		}
	}
	__syncthreads();
	if ( threadIdx.x < 6 )
	{
		float v = red[threadIdx.x][0];
		for ( int w = 1; w < TB / 64; ++w )
		{
			const float o = red[threadIdx.x][w];
			if ( threadIdx.x < 3 ? o < v : o > v ) v = o;
		}
		part[6 * blockIdx.x + threadIdx.x] = v;
	}
}

// one wave: folds the partial boxes, applies the rules for vertex 0 and the margin
__global__ __launch_bounds__( 64 ) void k_box_final( const rt_vertex *verts, uint64_t nV, const float *part, int nPart, GeomInfo *info )
{
	const float inf = __builtin_huge_valf();
	float mn[3] = { inf, inf, inf }, mx[3] = { -inf, -inf, -inf };
	for ( int b = threadIdx.x; b < nPart; b += 64 )
		for ( int a = 0; a < 3; ++a )
		{
			const float o = part[6 * b + a], q = part[6 * b + 3 + a];
			if ( o < mn[a] ) mn[a] = o;
			if ( q > mx[a] ) mx[a] = q;
		}
	for ( int a = 0; a < 3; ++a )
		for ( int off = 32; off; off >>= 1 )
		{
			const float o = __shfl_xor( mn[a], off ), q = __shfl_xor( mx[a], off );
			if ( o < mn[a] ) mn[a] = o;
			if ( q > mx[a] ) mx[a] = q;
		}
	if ( threadIdx.x != 0 )
		return;
	float first[3] = { 0.0f, 0.0f, 0.0f };
	if ( nV )
	{
		const rt_vector3 p = verts[0].pos;
		first[0] = p.x; first[1] = p.y; first[2] = p.z;
	}
	for ( int a = 0; a < 3; ++a )
	{
		float lo = mn[a], hi = mx[a];
		if ( nV == 0 ) lo = hi = 0.0f;
		else if ( first[a] != first[a] ) lo = hi = first[a];
		lo -= 1.0f;
		hi += 1.0f;
		const float c = ( lo + hi ) * 0.5f, e = ( hi - lo ) * 0.5f;
		info->box[a] = c;
		info->box[3 + a] = e;
	}
}

__global__ __launch_bounds__( TB ) void k_iota( uint32_t *refs, uint32_t *refNode, uint32_t n )
{
	const uint32_t i = blockIdx.x * TB + threadIdx.x;
	if ( i < n )
	{
		refs[i] = i;
		refNode[i] = 0u;
	}
}

// ---- the level loop's state lives in device memory --------------------------
// The host enqueues every level up to the depth limit without looking at a count:
// launches have fixed grids, every kernel reads its level's counts from here and
// returns when there is nothing to do (levels below the deepest one, or a work
// buffer turned out too small: then the host grows it and builds again).
struct LevelState
{
	uint32_t nodes, refs;       // nodes of this depth, triangle references held by them
	uint32_t nodeBase;          // nodes of all levels above (= level-order number of this level's first node)
	uint32_t leafBase;          // references in leaves of all levels above
};

enum : uint32_t { OVF_NONE = 0u, OVF_REFS = 1u, OVF_LEVEL_NODES = 2u, OVF_NODES = 3u, OVF_LEAF = 4u, OVF_LINKS = 5u };

struct BuildState
{
	LevelState lv[RT_HIP_MAX_TREE_DEPTH + 2];
	uint32_t overflow;          // OVF_*
	uint32_t need;              // the count that did not fit
	uint32_t nLevels, nodesTotal, leafRefs;     // written by the level that splits nothing
};

struct Caps { uint32_t refs, levelNodes, nodesTotal, leaf; };

// ---- node table of one level (one entry per node of the level) ---------------
enum : uint32_t { K_SPLIT = 1u, K_SORTED = 8u, K_DONE = 16u };     // (bits 1-2: split axis)

struct NodeTab
{
	uint4 *excl;           // over the nodes before this one: x inner nodes, y references handed to children, z references in leaves, w references handed to left children
	float *box;            // [6 * n]: centre xyz, half extent xyz
	uint32_t *start, *count;
	uint32_t *parent;      // index in the level above
	uint32_t *kind;        // K_SPLIT: inner node; axis; K_SORTED: candidate planes ascend (always, except for NaN / inverted boxes); K_DONE: split chosen by the kernel that counted
	float *mn, *delta;     // candidate plane k = mn + k * delta
	float *sep;            // chosen plane
	uint32_t *cntL, *cntR;
};
constexpr int NODE_TAB_WORDS = 4 + 6 + 9;

constexpr int HIST = 2 * PARTS + 2;      // per node: left histogram [33], right histogram [33] (sorted planes); else L[32], R[32] at 0 and 33
constexpr int HSLOTS = 64;               // nodes of one tile of references counted in LDS
constexpr int G_REF = 4096, G_NODE = 256;  // blocks of the per-reference / per-node part of a launch
constexpr int SPLIT_FINER = 8;           // k_split's blocks take an eighth of a k_children block's nodes each
static_assert( PARTS == 32, "the split search gives one plane to each lane of half a wave" );

__device__ inline float box_c( const float *b, int ax ) { return b[ax]; }
__device__ inline float box_e( const float *b, int ax ) { return b[3 + ax]; }

// contiguous share of n nodes for block blk of nBlocks, a multiple of TB long
__device__ inline uint32_t node_share( uint32_t n, uint32_t nBlocks ) { return ( ( n + nBlocks * TB - 1u ) / ( nBlocks * TB ) ) * TB; }

__device__ inline void share_of( uint32_t n, uint32_t per, uint32_t blk, uint32_t &b0, uint32_t &b1 )
{
	const uint64_t s = (uint64_t) blk * per;
	b0 = s < n ? (uint32_t) s : n;
	b1 = s + per < n ? (uint32_t) ( s + per ) : n;
}

// Contiguous share of a level's references for block blk of nBlocks, in tiles of TB.  Near the root a node holds so
// many references that its counts would be flushed by thousands of blocks, one atomic after the other on the same 66
// words: there a block takes more tiles (it keeps the counts of a node in LDS from tile to tile), so that a node
// is shared by about 512 blocks at most.
__device__ inline void ref_share( const LevelState &L, uint32_t nBlocks, uint32_t blk, uint32_t &b0, uint32_t &b1 )
{
	uint32_t tiles = ( L.refs + nBlocks * TB - 1u ) / ( nBlocks * TB );
	const uint32_t perNode = L.refs / ( L.nodes ? L.nodes : 1u ) / ( 512u * TB );
	if ( tiles < perNode ) tiles = perNode;
	share_of( L.refs, tiles * TB, blk, b0, b1 );
}

__device__ inline uint32_t wave_sum( uint32_t v )
{
	for ( int off = 32; off; off >>= 1 )
		v += (uint32_t) __shfl_xor( (int) v, off );
	return v;
}

// sum of N values over the block, returned to every thread (red: [TB / 64][N] words of LDS)
template <int N>
__device__ inline void block_sum( uint32_t ( &v )[N], uint32_t *red )
{
	for ( int k = 0; k < N; ++k )
		v[k] = wave_sum( v[k] );
	__syncthreads();
	if ( ( threadIdx.x & 63 ) == 0 )
		for ( int k = 0; k < N; ++k )
			red[( threadIdx.x >> 6 ) * N + k] = v[k];
	__syncthreads();
	for ( int k = 0; k < N; ++k )
	{
		uint32_t s = 0;
		for ( int w = 0; w < TB / 64; ++w )
			s += red[w * N + k];
		v[k] = s;
	}
}

// leaf (count <= maxPrims or depth == maxDepth) or inner node with its split axis:
// longest half extent, ties X > Y > Z (:945-957), and the candidate planes
__device__ inline void decide_node( const NodeTab &T, uint32_t i, const float *b, uint32_t cnt, int depth, int maxDepth, int maxPrims )
{
	if ( cnt <= (uint32_t) maxPrims || depth == maxDepth )
	{
		T.kind[i] = 0u;
		return;
	}
	float range = b[3];
	int ax = 0;
	if ( range < b[4] ) { range = b[4]; ax = 1; }
	if ( range < b[5] ) { range = b[5]; ax = 2; }
	const float c = box_c( b, ax ), e = box_e( b, ax );
	const float mn = c - e, mx = c + e;
	const float delta = ( mx - mn ) / (float) PARTS;
	bool sorted = true;
	float prev = 0.0f;
	for ( int k = 0; k < PARTS; ++k )
	{
		const float s = mn + k * delta;
		if ( k && !( prev <= s ) ) sorted = false;
		prev = s;
	}
	T.kind[i] = K_SPLIT | ( (uint32_t) ax << 1 ) | ( sorted ? K_SORTED : 0u );
	T.mn[i] = mn;
	T.delta[i] = delta;
}

// the root and the state of level 0 (the state has been zeroed: a level that was never reached holds nothing)
__global__ __launch_bounds__( 64 ) void k_root( BuildState *st, NodeTab T, const GeomInfo *geom, uint32_t nT, int maxDepth, int maxPrims )
{
	if ( threadIdx.x != 0 )
		return;
	float b[6];
	for ( int k = 0; k < 6; ++k )
		T.box[k] = b[k] = geom->box[k];
	T.start[0] = 0u;
	T.count[0] = nT;
	T.parent[0] = 0u;
	decide_node( T, 0u, b, nT, 0, maxDepth, maxPrims );
	st->lv[0].nodes = 1u;
	st->lv[0].refs = nT;
}

// SAH sweep of :591-857 on the counts of one node, by 32 lanes (lane l = plane l):
// the lowest index with the strictly smallest cost.  h: the node's histogram (LDS or HBM).
__device__ inline void sah_node( const NodeTab &T, uint32_t i, uint32_t kd, const uint32_t *h, uint32_t l )
{
	const float *b = T.box + 6ull * i;
	const int ax = (int) ( ( kd >> 1 ) & 3u );
	const bool sorted = ( kd & K_SORTED ) != 0u;
	const float c = box_c( b, ax ), e = box_e( b, ax );
	const float mn = T.mn[i], delta = T.delta[i];
	float eA, eB;
	if ( ax == 0 ) { eA = b[4]; eB = b[5]; }
	else if ( ax == 1 ) { eA = b[3]; eB = b[5]; }
	else { eA = b[3]; eB = b[4]; }
	const float area = eA * eB;
	// sorted planes: L[k] = histL[0 .. k], R[k] = histR[k + 1 .. 32]; else the counts themselves
	uint32_t Lk = h[l], Rk = sorted ? h[33u + l + 1u] : h[33u + l];
	if ( sorted )
		for ( uint32_t off = 1u; off < 32u; off <<= 1 )
		{
			const uint32_t up = (uint32_t) __shfl_up( (int) Lk, off, 32 ), dn = (uint32_t) __shfl_down( (int) Rk, off, 32 );
			if ( l >= off ) Lk += up;
			if ( l + off < 32u ) Rk += dn;
		}
	const float sep = mn + (int) l * delta;
	const float sah = ( area + ( sep - c + e ) * ( eA + eB ) ) * (float) Lk + ( area + ( c + e - sep ) * ( eA + eB ) ) * (float) Rk;
	// the reference keeps plane 0 unless a later cost is smaller (a NaN never is): plane 0 if its cost is a NaN,
	// else the lowest index among the smallest costs that are numbers
	const bool nan0 = __shfl( (int) ( sah != sah ), 0, 32 ) != 0;
	float v = sah != sah ? __builtin_huge_valf() : sah;
	uint32_t best = l;
	for ( int off = 16; off; off >>= 1 )
	{
		const float ov = __shfl_xor( v, off, 32 );
		const uint32_t ob = (uint32_t) __shfl_xor( (int) best, off, 32 );
		if ( ov < v || ( ov == v && ob < best ) ) { v = ov; best = ob; }
	}
	if ( nan0 ) best = 0u;
	if ( l == best )
	{
		T.sep[i] = sep;
		T.cntL[i] = Lk;
		T.cntR[i] = Rk;
	}
}

__device__ inline void hist_clear( uint32_t *h, uint32_t l )
{
	h[l] = 0u;
	h[l + 32u] = 0u;
	if ( l < 2u ) h[64u + l] = 0u;
}

// One triangle extent [a, b] against the 32 planes of node i of table T: into the node's slot of the LDS histograms
// when it has one (slot = node index - first node of the tile), else into its histogram in HBM.
__device__ inline void hist_add( const NodeTab &T, uint32_t i, uint32_t kd, uint32_t tri, const float *ext, uint32_t nT, uint32_t first,
	uint32_t *sh, uint32_t *hist )
{
	const uint32_t ax = ( kd >> 1 ) & 3u;
	const float a = ext[(size_t) ax * nT + tri], b = ext[( 3ull + ax ) * nT + tri];
	const float mn = T.mn[i], delta = T.delta[i];
	const uint32_t slot = i - first;
	const bool inLds = slot < (uint32_t) HSLOTS;
	uint32_t *hl = sh + ( inLds ? slot : 0u ) * HIST, *hg = hist + (size_t) HIST * i;
	if ( kd & K_SORTED )
	{
		// lo = number of planes at or below the triangle's minimum, hi = number of planes below its maximum
		int lo = 0, hi = 0;
		for ( int step = PARTS / 2; step; step >>= 1 )
		{
			if ( !( a < mn + ( lo + step - 1 ) * delta ) ) lo += step;
			if ( b > mn + ( hi + step - 1 ) * delta ) hi += step;
		}
		if ( lo < PARTS && !( a < mn + lo * delta ) ) ++lo;
		if ( hi < PARTS && b > mn + hi * delta ) ++hi;
		if ( inLds ) { atomicAdd( &hl[lo], 1u ); atomicAdd( &hl[33 + hi], 1u ); }
		else { atomicAdd( &hg[lo], 1u ); atomicAdd( &hg[33 + hi], 1u ); }
	}
	else
	{
		for ( int k = 0; k < PARTS; ++k )
		{
			const float s = mn + k * delta;
			if ( a < s ) { if ( inLds ) atomicAdd( &hl[k], 1u ); else atomicAdd( &hg[k], 1u ); }
			if ( b > s ) { if ( inLds ) atomicAdd( &hl[33 + k], 1u ); else atomicAdd( &hg[33 + k], 1u ); }
		}
	}
}

// After a tile of references has been counted: half a wave per node that has a slot in LDS (nodes first ..
// first + nSlots - 1 of table T).  A node all of whose references were in this tile gets its split chosen right
// here and never touches the histograms in HBM; the first `keep` slots stay as they are (their nodes go on in the
// next tile); the counts of the others are added to theirs in HBM.
template <class Complete>
__device__ inline void finish_slots( const NodeTab &T, uint32_t first, uint32_t nSlots, uint32_t keep, uint32_t *sh, uint32_t *hist, Complete complete )
{
	const uint32_t l = threadIdx.x & 31u;
	for ( uint32_t s = threadIdx.x >> 5; s < nSlots; s += TB / 32 )
	{
		const uint32_t i = first + s, kd = T.kind[i];
		if ( !( kd & K_SPLIT ) )
			continue;
		uint32_t *h = sh + s * HIST;
		if ( complete( i ) )
		{
			sah_node( T, i, kd, h, l );
			hist_clear( h, l );
			if ( l == 0u )
				T.kind[i] = kd | K_DONE;
		}
		else if ( s >= keep )
		{
			uint32_t *g = hist + (size_t) HIST * i;
			for ( uint32_t k = l; k < (uint32_t) HIST; k += 32u )
			{
				const uint32_t v = h[k];
				if ( v )
				{
					atomicAdd( &g[k], v );
					h[k] = 0u;
				}
			}
		}
	}
}

// Histogram of every triangle extent against the 32 planes of its node, for the root level (below it the
// partition counts for the level it fills, k_scatter).  References are ordered by node, a block takes a contiguous
// run of them in tiles of TB; the nodes a tile touches are counted in LDS (the first HSLOTS of them).
__global__ __launch_bounds__( TB ) void k_hist( const BuildState *st, int depth, NodeTab T, uint32_t *hist, const uint32_t *refs,
	const uint32_t *refNode, const float *ext, uint32_t nT )
{
	if ( st->overflow != OVF_NONE )
		return;
	const LevelState L = st->lv[depth];
	uint32_t b0, b1;
	ref_share( L, gridDim.x, blockIdx.x, b0, b1 );
	if ( b0 >= b1 )
		return;
	__shared__ uint32_t sh[HSLOTS * HIST];
	for ( int k = threadIdx.x; k < HSLOTS * HIST; k += TB )
		sh[k] = 0u;
	__syncthreads();
	for ( uint32_t j0 = b0; j0 < b1; j0 += TB )
	{
		const uint32_t tileEnd = j0 + TB < b1 ? j0 + TB : b1;
		const uint32_t node0 = refNode[j0], nodeLast = refNode[tileEnd - 1u];
		const uint32_t j = j0 + threadIdx.x;
		if ( j < tileEnd )
		{
			const uint32_t i = refNode[j], kd = T.kind[i];
			if ( kd & K_SPLIT )
				hist_add( T, i, kd, refs[j], ext, nT, node0, sh, hist );
		}
		__syncthreads();
		const uint32_t keep = tileEnd < b1 && refNode[tileEnd] == node0 ? 1u : 0u;
		const uint32_t span = nodeLast - node0 + 1u;
		finish_slots( T, node0, span < (uint32_t) HSLOTS ? span : (uint32_t) HSLOTS, keep, sh, hist,
			[&]( uint32_t i ) { const uint32_t first = T.start[i]; return first >= j0 && first + T.count[i] <= tileEnd; } );
		__syncthreads();
	}
}

// The nodes whose split the counting kernel could not choose (they span tiles): from the histograms in HBM, which
// are left zeroed for the next level; then the sums of this block's nodes for k_children's scan.  A block takes one
// SPLIT_FINER-th of a k_children block's nodes (the splits are a chain of dependent loads per node).
__global__ __launch_bounds__( TB ) void k_split( const BuildState *st, int depth, NodeTab T, uint32_t *hist, uint4 *blockSums )
{
	if ( st->overflow != OVF_NONE )
		return;
	const uint32_t n = st->lv[depth].nodes;
	if ( n == 0u )
		return;
	__shared__ uint32_t red[( TB / 64 ) * 4];
	__shared__ uint32_t todo[TB], nTodo;
	uint32_t b0, b1;
	share_of( n, node_share( n, G_NODE ) / SPLIT_FINER, blockIdx.x, b0, b1 );
	const uint32_t l = threadIdx.x & 31u;
	uint32_t acc[4] = { 0u, 0u, 0u, 0u };
	for ( uint32_t i0 = b0; i0 < b1; i0 += TB )          // (b0 == b1: nothing but the zero sums)
	{
		if ( threadIdx.x == 0 )
			nTodo = 0u;
		__syncthreads();
		const uint32_t i = i0 + threadIdx.x;
		uint32_t kd = 0u;
		if ( i < b1 )
		{
			kd = T.kind[i];
			if ( ( kd & ( K_SPLIT | K_DONE ) ) == K_SPLIT )
				todo[atomicAdd( &nTodo, 1u )] = i;
		}
		__syncthreads();
		const uint32_t m = nTodo;
		for ( uint32_t q = threadIdx.x >> 5; q < m; q += TB / 32 )
		{
			const uint32_t ni = todo[q];
			uint32_t *h = hist + (size_t) HIST * ni;
			sah_node( T, ni, T.kind[ni], h, l );
			hist_clear( h, l );
		}
		__syncthreads();
		if ( i < b1 )
		{
			if ( kd & K_SPLIT )
			{
				const uint32_t cl = T.cntL[i], cr = T.cntR[i];
				acc[0] += 1u;
				acc[1] += cl + cr;
				acc[3] += cl;
			}
			else
				acc[2] += T.count[i];
		}
	}
	block_sum( acc, red );
	if ( threadIdx.x == 0 )
		blockSums[blockIdx.x] = make_uint4( acc[0], acc[1], acc[2], acc[3] );
}

// does reference j go to the left / right child of its node (min < sep / max > sep, :971-1029)
struct RefSide
{
	uint32_t node, kind, tri;
	bool left, right;
};

__device__ inline RefSide ref_side( const NodeTab &T, const uint32_t *refs, const uint32_t *refNode, const float *ext, uint32_t nT, uint32_t j )
{
	RefSide r;
	r.node = refNode[j];
	r.kind = T.kind[r.node];
	r.tri = refs[j];
	r.left = r.right = false;
	if ( r.kind & K_SPLIT )
	{
		const uint32_t ax = ( r.kind >> 1 ) & 3u;
		const float sep = T.sep[r.node];
		r.left = ext[(size_t) ax * nT + r.tri] < sep;
		r.right = ext[( 3ull + ax ) * nT + r.tri] > sep;
	}
	return r;
}

// Two independent pieces of work in one launch (both need nothing but k_split's results).
// Blocks [0, G_NODE): number the children (a scan over the level's nodes, in two steps: the sums of k_split's blocks,
// then inside the block), write the child boxes - slab split at sep (:1034-1073) -, decide leaf / inner node for
// them, write this level's records of the tree in level order, and the next level's state.  Every block works out the
// level's totals for itself, so all of them agree whether the next level fits the work buffers.
// Blocks [G_NODE, G_NODE + G_REF): references going left / right in every block's share of the level (the first
// step of the partition's scan).
__global__ __launch_bounds__( TB ) void k_children_count( BuildState *st, int depth, NodeTab T, NodeTab N, const uint4 *blockSums, uint2 *info,
	Caps caps, int maxDepth, int maxPrims, const uint32_t *refs, const uint32_t *refNode, const float *ext, uint32_t nT, uint2 *refSums )
{
	if ( st->overflow != OVF_NONE )
		return;
	const LevelState L = st->lv[depth];
	const uint32_t n = L.nodes;
	if ( n == 0u )
		return;
	__shared__ uint32_t red[( TB / 64 ) * 8];
	__shared__ uint32_t waveTot[( TB / 64 ) * 4];
	if ( blockIdx.x >= (uint32_t) G_NODE )
	{
		uint32_t b0, b1;
		ref_share( L, G_REF, blockIdx.x - G_NODE, b0, b1 );
		uint32_t acc[2] = { 0u, 0u };
		for ( uint32_t j = b0 + threadIdx.x; j < b1; j += TB )
		{
			const RefSide r = ref_side( T, refs, refNode, ext, nT, j );
			acc[0] += r.left ? 1u : 0u;
			acc[1] += r.right ? 1u : 0u;
		}
		block_sum( acc, red );
		if ( threadIdx.x == 0 )
			refSums[blockIdx.x - G_NODE] = make_uint2( acc[0], acc[1] );
		return;
	}
	uint32_t pt[8] = { 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u };      // sums of the nodes before this block's, of all nodes
	for ( uint32_t b = threadIdx.x; b < (uint32_t) ( G_NODE * SPLIT_FINER ); b += TB )
	{
		const uint4 v = blockSums[b];
		if ( b < blockIdx.x * SPLIT_FINER ) { pt[0] += v.x; pt[1] += v.y; pt[2] += v.z; pt[3] += v.w; }
		pt[4] += v.x; pt[5] += v.y; pt[6] += v.z; pt[7] += v.w;
	}
	block_sum( pt, red );
	const uint32_t splits = pt[4], childRefs = pt[5], levelLeafRefs = pt[6];
	const uint64_t nodesAfter = (uint64_t) L.nodeBase + n + 2ull * splits, leafAfter = (uint64_t) L.leafBase + levelLeafRefs;
	uint32_t ovf = OVF_NONE, need = 0u;
	if ( nodesAfter >= ( 1ull << 30 ) || leafAfter + childRefs >= ( 1ull << 30 ) ) ovf = OVF_LINKS;      // 30-bit links of the 8-byte node
	else if ( 2ull * splits > caps.levelNodes ) { ovf = OVF_LEVEL_NODES; need = 2u * splits; }
	else if ( childRefs > caps.refs ) { ovf = OVF_REFS; need = childRefs; }
	else if ( nodesAfter > caps.nodesTotal ) { ovf = OVF_NODES; need = (uint32_t) nodesAfter; }
	else if ( leafAfter > caps.leaf ) { ovf = OVF_LEAF; need = (uint32_t) leafAfter; }
	if ( blockIdx.x == 0 && threadIdx.x == 0 )
	{
		if ( ovf != OVF_NONE )
		{
			st->overflow = ovf;
			st->need = need;
		}
		else
		{
			LevelState nx;
			nx.nodes = 2u * splits;
			nx.refs = childRefs;
			nx.nodeBase = L.nodeBase + n;
			nx.leafBase = (uint32_t) leafAfter;
			st->lv[depth + 1] = nx;
			if ( splits == 0u )
			{
				st->nLevels = (uint32_t) depth + 1u;
				st->nodesTotal = L.nodeBase + n;
				st->leafRefs = (uint32_t) leafAfter;
			}
		}
	}
	if ( ovf != OVF_NONE )
		return;
	uint32_t b0, b1;
	share_of( n, node_share( n, G_NODE ), blockIdx.x, b0, b1 );
	uint32_t run[4] = { pt[0], pt[1], pt[2], pt[3] };
	const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
	const uint32_t nextBase = L.nodeBase + n;
	for ( uint32_t i0 = b0; i0 < b1; i0 += TB )
	{
		const uint32_t i = i0 + threadIdx.x;
		const bool valid = i < b1;
		uint32_t kd = 0u, own[4] = { 0u, 0u, 0u, 0u }, cnt = 0u, cl = 0u, cr = 0u;
		if ( valid )
		{
			kd = T.kind[i];
			cnt = T.count[i];
			if ( kd & K_SPLIT )
			{
				cl = T.cntL[i];
				cr = T.cntR[i];
				own[0] = 1u; own[1] = cl + cr; own[3] = cl;
			}
			else
				own[2] = cnt;
		}
		// inclusive scan inside the wave, wave totals through LDS
		uint32_t inc[4];
		for ( int k = 0; k < 4; ++k )
		{
			uint32_t v = own[k];
			for ( uint32_t off = 1u; off < 64u; off <<= 1 )
			{
				const uint32_t up = (uint32_t) __shfl_up( (int) v, off );
				if ( lane >= off ) v += up;
			}
			inc[k] = v;
		}
		__syncthreads();          // (the previous tile's wave totals have been read)
		if ( lane == 63u )
			for ( int k = 0; k < 4; ++k )
				waveTot[w * 4 + k] = inc[k];
		__syncthreads();
		uint32_t ex[4];
		for ( int k = 0; k < 4; ++k )
		{
			uint32_t before = run[k], all = 0u;
			for ( uint32_t q = 0; q < TB / 64; ++q )
			{
				const uint32_t t = waveTot[q * 4 + k];
				if ( q < w ) before += t;
				all += t;
			}
			ex[k] = before + inc[k] - own[k];
			run[k] += all;
		}
		if ( !valid )
			continue;
		T.excl[i] = make_uint4( ex[0], ex[1], ex[2], ex[3] );
		if ( !( kd & K_SPLIT ) )
		{
			info[L.nodeBase + i] = make_uint2( L.leafBase + ex[2], ( cnt << 2 ) | 3u );
			continue;
		}
		float b[6], bl[6], br[6];
		for ( int k = 0; k < 6; ++k )
			bl[k] = br[k] = b[k] = T.box[6ull * i + k];
		const int ax = (int) ( ( kd >> 1 ) & 3u );
		const float sep = T.sep[i];
		const uint32_t c = 2u * ex[0];
		bl[ax] = 0.5f * ( box_c( b, ax ) - box_e( b, ax ) + sep );
		bl[3 + ax] = 0.5f * ( sep - box_c( b, ax ) + box_e( b, ax ) );
		br[ax] = 0.5f * ( box_c( b, ax ) + box_e( b, ax ) + sep );
		br[3 + ax] = 0.5f * ( box_c( b, ax ) + box_e( b, ax ) - sep );
		for ( int k = 0; k < 6; ++k )
		{
			N.box[6ull * c + k] = bl[k];
			N.box[6ull * c + 6 + k] = br[k];
		}
		N.start[c] = ex[1];
		N.count[c] = cl;
		N.start[c + 1] = ex[1] + cl;
		N.count[c + 1] = cr;
		N.parent[c] = N.parent[c + 1] = i;
		decide_node( N, c, bl, cl, depth + 1, maxDepth, maxPrims );
		decide_node( N, c + 1u, br, cr, depth + 1, maxDepth, maxPrims );
		info[L.nodeBase + i] = make_uint2( __float_as_uint( sep ), ( ( nextBase + c ) << 2 ) | (uint32_t) ax );
	}
}

// Stable partition (:971-1029): the k-th reference of a node that goes left becomes the k-th reference of its left
// child; references of leaves move to the leaf pool.  A reference's rank among those of its node that go left is its
// rank among all of the level that go left, less the left counts of the nodes before its own (NodeTab::excl).
// A reference that goes to a child is counted against that child's planes on the way (the children of the nodes of a
// tile are neighbours in the next level's table N: the same LDS histograms and in-tile splits as in k_hist, a child
// being complete when its parent's references all lie in the tile), so the next level starts with its counts taken.
__global__ __launch_bounds__( TB ) void k_scatter( const BuildState *st, int depth, NodeTab T, NodeTab N, const uint32_t *refs, const uint32_t *refNode,
	const float *ext, uint32_t nT, const uint2 *blockSums, uint32_t *nextRefs, uint32_t *nextRefNode, uint32_t *leafPool,
	uint32_t *hist, uint32_t *lostInPlane )
{
	if ( st->overflow != OVF_NONE )
		return;
	const LevelState L = st->lv[depth];
	uint32_t b0, b1;
	ref_share( L, gridDim.x, blockIdx.x, b0, b1 );
	if ( b0 >= b1 )
		return;
	__shared__ uint32_t red[( TB / 64 ) * 2];
	__shared__ uint32_t waveTot[( TB / 64 ) * 2];
	__shared__ uint32_t sh[HSLOTS * HIST];
	for ( int k = threadIdx.x; k < HSLOTS * HIST; k += TB )
		sh[k] = 0u;
	uint32_t run[2] = { 0u, 0u };
	for ( uint32_t b = threadIdx.x; b < blockIdx.x; b += TB )
	{
		const uint2 v = blockSums[b];
		run[0] += v.x;
		run[1] += v.y;
	}
	block_sum( run, red );
	const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
	const unsigned long long below = ( 1ull << lane ) - 1ull;
	for ( uint32_t j0 = b0; j0 < b1; j0 += TB )
	{
		const uint32_t tileEnd = j0 + TB < b1 ? j0 + TB : b1;
		const uint32_t node0 = refNode[j0], nodeLast = refNode[tileEnd - 1u];
		const uint32_t j = j0 + threadIdx.x;
		const bool valid = j < tileEnd;
		RefSide r;
		r.node = r.kind = r.tri = 0u;
		r.left = r.right = false;
		if ( valid )
			r = ref_side( T, refs, refNode, ext, nT, j );
		const unsigned long long mL = __ballot( r.left ), mR = __ballot( r.right );
		if ( lane == 0u )
		{
			waveTot[w * 2] = (uint32_t) __popcll( mL );
			waveTot[w * 2 + 1] = (uint32_t) __popcll( mR );
		}
		const uint32_t kd0 = T.kind[node0], kdLast = T.kind[nodeLast];
		const uint32_t c0 = 2u * T.excl[node0].x, cEnd = 2u * ( T.excl[nodeLast].x + ( kdLast & K_SPLIT ) );
		__syncthreads();
		uint32_t gl = run[0], gr = run[1];
		for ( uint32_t q = 0; q < TB / 64; ++q )
		{
			const uint32_t tl = waveTot[q * 2], tr = waveTot[q * 2 + 1];
			if ( q < w ) { gl += tl; gr += tr; }
			run[0] += tl;
			run[1] += tr;
		}
		gl += (uint32_t) __popcll( mL & below );
		gr += (uint32_t) __popcll( mR & below );
		if ( valid )
		{
			const uint4 ex = T.excl[r.node];
			if ( !( r.kind & K_SPLIT ) )
				leafPool[L.leafBase + ex.z + ( j - T.start[r.node] )] = r.tri;
			else
			{
				const uint32_t c = 2u * ex.x;
				if ( !r.left && !r.right )
					atomicAdd( lostInPlane, 1u );             // (rare: a triangle flat in the split axis that lies in the plane)
				if ( r.left )
				{
					const uint32_t p = ex.y + ( gl - ex.w );
					nextRefs[p] = r.tri;
					nextRefNode[p] = c;
					const uint32_t kdc = N.kind[c];
					if ( kdc & K_SPLIT )
						hist_add( N, c, kdc, r.tri, ext, nT, c0, sh, hist );
				}
				if ( r.right )
				{
					const uint32_t p = ex.y + T.cntL[r.node] + ( gr - ( ex.y - ex.w ) );
					nextRefs[p] = r.tri;
					nextRefNode[p] = c + 1u;
					const uint32_t kdc = N.kind[c + 1u];
					if ( kdc & K_SPLIT )
						hist_add( N, c + 1u, kdc, r.tri, ext, nT, c0, sh, hist );
				}
			}
		}
		__syncthreads();
		// (the first node of the tile goes on in the next one: its two children keep their counts in LDS)
		const uint32_t keep = tileEnd < b1 && refNode[tileEnd] == node0 && ( kd0 & K_SPLIT ) ? 2u : 0u;
		const uint32_t span = cEnd - c0;
		finish_slots( N, c0, span < (uint32_t) HSLOTS ? span : (uint32_t) HSLOTS, keep, sh, hist,
			[&]( uint32_t cn ) { const uint32_t par = N.parent[cn], first = T.start[par]; return first >= j0 && first + T.count[par] <= tileEnd; } );
		__syncthreads();
	}
}

// ---- finished tree: sizes, positions, output --------------------------------
__global__ __launch_bounds__( TB ) void k_sizes( const uint2 *info, uint32_t first, uint32_t n, uint32_t *subNodes, uint32_t *subRefs, uint32_t *subDepth )
{
	const uint32_t k = blockIdx.x * TB + threadIdx.x;
	if ( k >= n )
		return;
	const uint32_t id = first + k;
	const uint2 v = info[id];
	if ( ( v.y & 3u ) == 3u )
	{
		subNodes[id] = 1u;
		subRefs[id] = v.y >> 2;
		subDepth[id] = 0u;
		return;
	}
	const uint32_t l = v.y >> 2;
	subNodes[id] = 1u + subNodes[l] + subNodes[l + 1];
	subRefs[id] = subRefs[l] + subRefs[l + 1];
	subDepth[id] = 1u + ( subDepth[l] > subDepth[l + 1] ? subDepth[l] : subDepth[l + 1] );
}

// pre-order numbering, left child right behind its parent (:860-912)
__global__ __launch_bounds__( TB ) void k_positions( const uint2 *info, uint32_t first, uint32_t n, const uint32_t *subNodes, const uint32_t *subRefs,
	uint32_t *pos, uint32_t *refPos )
{
	const uint32_t k = blockIdx.x * TB + threadIdx.x;
	if ( k >= n )
		return;
	const uint32_t id = first + k;
	const uint2 v = info[id];
	if ( ( v.y & 3u ) == 3u )
		return;
	const uint32_t l = v.y >> 2;
	pos[l] = pos[id] + 1u;
	pos[l + 1] = pos[id] + 1u + subNodes[l];
	refPos[l] = refPos[id];
	refPos[l + 1] = refPos[id] + subRefs[l];
}

__global__ __launch_bounds__( TB ) void k_emit( const uint2 *info, uint32_t n, const uint32_t *pos, const uint32_t *refPos, const uint32_t *leafPool,
	rt_hip_kdnode *nodes, uint32_t *order )
{
	const uint32_t id = blockIdx.x * TB + threadIdx.x;
	if ( id >= n )
		return;
	const uint2 v = info[id];
	rt_hip_kdnode o;
	if ( ( v.y & 3u ) == 3u )
	{
		const uint32_t cnt = v.y >> 2, dst = refPos[id];
		o.a = dst;
		o.b = v.y;
		for ( uint32_t k = 0; k < cnt; ++k )
			order[dst + k] = leafPool[v.x + k];
	}
	else
	{
		o.a = v.x;
		o.b = ( pos[( v.y >> 2 ) + 1u] << 2 ) | ( v.y & 3u );
	}
	nodes[pos[id]] = o;
}

__global__ __launch_bounds__( TB ) void k_records( const rt_vertex *verts, const rt_triangle *tris, const uint32_t *order, uint32_t nRecs, rt_hip_trirec *recs )
{
	const uint32_t i = blockIdx.x * TB + threadIdx.x;
	if ( i > nRecs )
		return;
	rt_hip_trirec r;
	memset( &r, 0, sizeof( r ) );
	if ( i < nRecs )              // entry nRecs is the zeroed spare record the look-ahead load may touch
	{
		const rt_triangle t = tris[order[i]];
		const rt_vector3 p0 = verts[t.pV0].pos, p1 = verts[t.pV1].pos, p2 = verts[t.pV2].pos;
		r.nv0x = -p0.x; r.nv0y = -p0.y; r.nv0z = -p0.z;
		r.e1x = p1.x - p0.x; r.e1y = p1.y - p0.y; r.e1z = p1.z - p0.z;
		r.e2x = p2.x - p0.x; r.e2y = p2.y - p0.y; r.e2z = p2.z - p0.z;
		r.tri = order[i];
		r.mat = (uint32_t) t.mat;
	}
	recs[i] = r;
}

__global__ __launch_bounds__( TB ) void k_shade_records( const rt_vertex *verts, const rt_triangle *tris, uint32_t nT, rt_hip_trishade *shade )
{
	const uint32_t i = blockIdx.x * TB + threadIdx.x;
	if ( i >= nT )
		return;
	const rt_triangle t = tris[i];
	const rt_vector3 n0 = verts[t.pV0].norm, n1 = verts[t.pV1].norm, n2 = verts[t.pV2].norm;
	rt_hip_trishade s;
	s.n0[0] = n0.x; s.n0[1] = n0.y; s.n0[2] = n0.z;
	s.n1[0] = n1.x; s.n1[1] = n1.y; s.n1[2] = n1.z;
	s.n2[0] = n2.x; s.n2[1] = n2.y; s.n2[2] = n2.z;
	s.mat = (uint32_t) t.mat;
	s._pad[0] = s._pad[1] = 0;
	shade[i] = s;
}

// ---- pruned leaf lists (rt_kdtree_build_opts.pruneLeaves, see rt_kdtree.h) ----
// Does the triangle touch the cell grown by `pad`?  Separating axes: the cell's
// three normals, the triangle's normal, the nine edge cross products; "yes" in
// every doubtful case (the same test as the host builder's).
__device__ bool tri_touches_cell( const double q[3][3], const float lo[3], const float hi[3], double pad )
{
	double bc[3], bh[3], a[3][3];
	for ( int k = 0; k < 3; ++k )
	{
		bc[k] = 0.5 * ( (double) lo[k] + (double) hi[k] );
		bh[k] = 0.5 * ( (double) hi[k] - (double) lo[k] ) + pad;
		for ( int v = 0; v < 3; ++v )
			a[v][k] = q[v][k] - bc[k];
	}
	for ( int k = 0; k < 3; ++k )
	{
		const double mn = fmin( a[0][k], fmin( a[1][k], a[2][k] ) ), mx = fmax( a[0][k], fmax( a[1][k], a[2][k] ) );
		const bool nan = a[0][k] != a[0][k] || a[1][k] != a[1][k] || a[2][k] != a[2][k];
		if ( !( mn <= bh[k] && mx >= -bh[k] ) )
			return nan;
		if ( nan )
			return true;
	}
	double e[3][3];
	for ( int k = 0; k < 3; ++k )
	{
		e[0][k] = a[1][k] - a[0][k];
		e[1][k] = a[2][k] - a[1][k];
		e[2][k] = a[0][k] - a[2][k];
	}
	{
		const double n[3] = { e[0][1] * -e[2][2] - e[0][2] * -e[2][1], e[0][2] * -e[2][0] - e[0][0] * -e[2][2], e[0][0] * -e[2][1] - e[0][1] * -e[2][0] };
		const double d = n[0] * a[0][0] + n[1] * a[0][1] + n[2] * a[0][2];
		const double r = fabs( n[0] ) * bh[0] + fabs( n[1] ) * bh[1] + fabs( n[2] ) * bh[2];
		if ( fabs( d ) > r )
			return false;
	}
	for ( int f = 0; f < 3; ++f )
		for ( int k = 0; k < 3; ++k )
		{
			const int k1 = ( k + 1 ) % 3, k2 = ( k + 2 ) % 3;
			const double ax1 = -e[f][k2], ax2 = e[f][k1];
			const double p0 = ax1 * a[0][k1] + ax2 * a[0][k2], p1 = ax1 * a[1][k1] + ax2 * a[1][k2], p2 = ax1 * a[2][k1] + ax2 * a[2][k2];
			const double r = fabs( ax1 ) * bh[k1] + fabs( ax2 ) * bh[k2];
			if ( fmin( p0, fmin( p1, p2 ) ) > r || fmax( p0, fmax( p1, p2 ) ) < -r )
				return false;
		}
	return true;
}

// One thread per leaf of the finished tree: finds the leaf's cell by walking down from the root
// (pre-order: the left sub-tree of node i is [i + 1, right), the right one starts at `right`), moves
// the records of the triangles that touch the cell to the front of the leaf's range, in order, and
// lowers the leaf's count.  The records behind them stay where they are, unused.  Nothing is done
// when the build dropped a reference (see rt_kdtree.h: such a tree keeps the reference's lists).
__global__ __launch_bounds__( TB ) void k_prune_leaves( rt_hip_kdnode *nodes, uint32_t nNodes, rt_hip_trirec *recs, const GeomInfo *geom, float margin )
{
	const uint32_t leaf = blockIdx.x * TB + threadIdx.x;
	if ( leaf >= nNodes || geom->lostInPlane != 0u )
		return;
	const rt_hip_kdnode me = nodes[leaf];
	if ( ( me.b & 3u ) != 3u || ( me.b >> 2 ) == 0u )
		return;
	float lo[3], hi[3];
	double size = 0.0;
	for ( int k = 0; k < 3; ++k )
	{
		lo[k] = geom->box[k] - geom->box[3 + k];
		hi[k] = geom->box[k] + geom->box[3 + k];
		size = fmax( size, fabs( (double) geom->box[k] ) + fabs( (double) geom->box[3 + k] ) );
	}
	for ( uint32_t i = 0u; i != leaf; )
	{
		const rt_hip_kdnode n = nodes[i];
		const uint32_t ax = n.b & 3u, right = n.b >> 2;
		const float sep = __uint_as_float( n.a );
		if ( leaf < right )
		{
			if ( ax == 0u ) hi[0] = sep; else if ( ax == 1u ) hi[1] = sep; else hi[2] = sep;
			i = i + 1u;
		}
		else
		{
			if ( ax == 0u ) lo[0] = sep; else if ( ax == 1u ) lo[1] = sep; else lo[2] = sep;
			i = right;
		}
	}
	const double pad = (double) margin * size;
	const uint32_t first = me.a, count = me.b >> 2;
	uint32_t kept = 0u;
	for ( uint32_t j = 0u; j < count; ++j )
	{
		const rt_hip_trirec r = recs[first + j];
		// (the record holds -v0 and the rounded differences v1 - v0, v2 - v0: v1, v2 are off by half a unit in the last place at most)
		double q[3][3];
		q[0][0] = -(double) r.nv0x; q[0][1] = -(double) r.nv0y; q[0][2] = -(double) r.nv0z;
		q[1][0] = q[0][0] + (double) r.e1x; q[1][1] = q[0][1] + (double) r.e1y; q[1][2] = q[0][2] + (double) r.e1z;
		q[2][0] = q[0][0] + (double) r.e2x; q[2][1] = q[0][1] + (double) r.e2y; q[2][2] = q[0][2] + (double) r.e2z;
		if ( tri_touches_cell( q, lo, hi, pad ) )
		{
			if ( kept != j )
				recs[first + kept] = r;
			++kept;
		}
	}
	if ( kept != count )
		nodes[leaf].b = ( kept << 2 ) | 3u;
}

// where every leaf's complete list lies once a copy of the record array has been put behind it (`shift` records on)
__global__ __launch_bounds__( TB ) void k_leaf_full( const rt_hip_kdnode *nodes, uint32_t nNodes, uint32_t shift, rt_hip_leaf_full *full )
{
	const uint32_t i = blockIdx.x * TB + threadIdx.x;
	if ( i >= nNodes )
		return;
	const rt_hip_kdnode n = nodes[i];
	rt_hip_leaf_full f = { 0u, 0u };
	if ( ( n.b & 3u ) == 3u )
	{
		f.first = n.a + shift;
		f.count = n.b >> 2;
	}
	full[i] = f;
}

inline unsigned grid_of( size_t n ) { return (unsigned) ( ( n + TB - 1 ) / TB ); }

struct Scratch
{
	DVec verts, tris, ext, refsA, refsB, tabA, tabB, hist, sums, info, leafPool, sub, order, geom, state;
	Caps caps = { 0u, 0u, 0u, 0u };
	bool histClean = false;      // every entry of `hist` is zero (each level leaves it so; a build that failed half-way may not)
	void *pinned = nullptr;      // BuildState + GeomInfo as the host reads them
	void release()
	{
		for ( DVec *v : { &verts, &tris, &ext, &refsA, &refsB, &tabA, &tabB, &hist, &sums, &info, &leafPool, &sub, &order, &geom, &state } )
			v->release();
		if ( pinned ) (void) hipHostFree( pinned );
		pinned = nullptr;
		caps = Caps{ 0u, 0u, 0u, 0u };
		histClean = false;
	}
};
thread_local Scratch g_scratch;
thread_local int g_scratchDev = -1;

// carve a level's node table out of one allocation
int node_tab( DVec &buf, size_t n, hipStream_t s, NodeTab &t )
{
	if ( int rc = buf.reserve( n * NODE_TAB_WORDS * sizeof( uint32_t ), s ) ) return rc;
	uint32_t *p = buf.as<uint32_t>();
	t.excl = (uint4 *) p; p += 4 * n;          // first: 16-byte aligned
	t.box = (float *) p; p += 6 * n;
	t.start = p; p += n;
	t.count = p; p += n;
	t.parent = p; p += n;
	t.kind = p; p += n;
	t.mn = (float *) p; p += n;
	t.delta = (float *) p; p += n;
	t.sep = (float *) p; p += n;
	t.cntL = p; p += n;
	t.cntR = p; p += n;
	return 0;
}

inline uint32_t cap_of( uint64_t want )
{
	return (uint32_t) ( want < ( 1ull << 30 ) ? want : ( 1ull << 30 ) );
}

} // namespace

extern "C" const char *rt_kdtree_device_last_error( void ) { return g_gpu_build_err; }

extern "C" void rt_kdtree_device_trim( void )
{
	if ( g_scratchDev >= 0 )
	{
		int cur = 0;
		if ( hipGetDevice( &cur ) == hipSuccess && hipSetDevice( g_scratchDev ) == hipSuccess )
		{
			g_scratch.release();
			(void) hipSetDevice( cur );
		}
		g_scratchDev = -1;
	}
}

extern "C" void rt_kdtree_device_release( rt_kdtree_device *t )
{
	if ( !t )
		return;
	if ( t->nodes ) (void) hipFree( t->nodes );
	if ( t->recs ) (void) hipFree( t->recs );
	if ( t->shade ) (void) hipFree( t->shade );
	if ( t->leafFull ) (void) hipFree( t->leafFull );
	memset( t, 0, sizeof( *t ) );
}

extern "C" int rt_kdtree_build_device( const rt_vertex *verts, const rt_triangle *tris, uint64_t vertexCount,
	uint64_t trianglesCount, int maxDepth, int maxPrims, void *hipStream, rt_kdtree_device *out )
{
	rt_kdtree_build_opts o;
	memset( &o, 0, sizeof( o ) );
	o.maxDepth = maxDepth;
	o.maxPrims = maxPrims;
	return rt_kdtree_build_device_ex( verts, tris, vertexCount, trianglesCount, &o, hipStream, out );
}

extern "C" int rt_kdtree_build_device_ex( const rt_vertex *verts, const rt_triangle *tris, uint64_t vertexCount,
	uint64_t trianglesCount, const rt_kdtree_build_opts *opts, void *hipStream, rt_kdtree_device *out )
{
	g_gpu_build_err[0] = 0;
	if ( !opts )
		return 1;
	if ( opts->sahStop > 0.0f || opts->allAxes )
		return 2;          // (host builder only)
	const int maxDepth = opts->maxDepth, maxPrims = opts->maxPrims;
	if ( !out || ( vertexCount && !verts ) || ( trianglesCount && !tris ) )
		return 1;
	if ( trianglesCount >= ( 1ull << 30 ) || maxDepth < 0 || maxDepth > RT_HIP_MAX_TREE_DEPTH || maxPrims < 1 )
		return 2;
	memset( out, 0, sizeof( *out ) );
	hipStream_t s = (hipStream_t) hipStream;
	const uint32_t nT = (uint32_t) trianglesCount;

	// Work buffers are kept between builds of one thread on one device (a frame
	// loop rebuilds a tree of about the same size every frame; allocating and
	// freeing them took as long as all the kernels); rt_kdtree_device_trim frees them.
	int dev = 0;
	GCHK( hipGetDevice( &dev ) );
	if ( g_scratchDev != dev )
	{
		g_scratch.release();
		g_scratchDev = dev;
	}
	const bool TT = getenv( "RT_KD_TIMING" ) != nullptr;
	auto t0_ = std::chrono::steady_clock::now();
	auto stamp = [&]( const char *what ) { if ( TT ) { fprintf( stderr, "  %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>( std::chrono::steady_clock::now() - t0_ ).count() ); } };
	Scratch &S = g_scratch;
	if ( !S.pinned )
		GCHK( hipHostMalloc( &S.pinned, sizeof( BuildState ) + sizeof( GeomInfo ), hipHostMallocDefault ) );
	BuildState *hState = (BuildState *) S.pinned;
	GeomInfo *hGeom = (GeomInfo *) ( hState + 1 );

	// inputs; the index check, the highest material slot and the scene box are
	// taken from the device copy (a host pass over 1 M triangles and their
	// vertices costs 2 ms, more than a quarter of the whole build)
	if ( int rc = S.verts.reserve( sizeof( rt_vertex ) * ( vertexCount ? vertexCount : 1 ), s ) ) return rc;
	if ( int rc = S.tris.reserve( sizeof( rt_triangle ) * ( nT ? nT : 1 ), s ) ) return rc;
	if ( vertexCount ) GCHK( hipMemcpyAsync( S.verts.p, verts, sizeof( rt_vertex ) * vertexCount, hipMemcpyHostToDevice, s ) );
	if ( nT ) GCHK( hipMemcpyAsync( S.tris.p, tris, sizeof( rt_triangle ) * nT, hipMemcpyHostToDevice, s ) );
	stamp( "input copies enqueued" );
	if ( TT ) { (void) hipStreamSynchronize( s ); stamp( "input copies done" ); }
	if ( int rc = S.ext.reserve( 6 * sizeof( float ) * ( nT ? nT : 1 ), s ) ) return rc;
	if ( int rc = S.geom.reserve( sizeof( GeomInfo ) + 6 * sizeof( float ) * BOX_BLOCKS, s ) ) return rc;
	if ( int rc = S.state.reserve( sizeof( BuildState ), s ) ) return rc;
	if ( int rc = S.sums.reserve( sizeof( uint4 ) * G_NODE * SPLIT_FINER + sizeof( uint2 ) * G_REF, s ) ) return rc;
	GeomInfo *dGeom = S.geom.as<GeomInfo>();
	BuildState *dState = S.state.as<BuildState>();
	uint4 *nodeSums = S.sums.as<uint4>();
	uint2 *refSums = (uint2 *) ( nodeSums + G_NODE * SPLIT_FINER );
	float *boxPart = (float *) ( dGeom + 1 );
	GCHK( hipMemsetAsync( dGeom, 0, sizeof( GeomInfo ), s ) );
	if ( nT ) LAUNCH( k_extents, dim3( grid_of( nT ) ), dim3( TB ), 0, s, S.verts.as<rt_vertex>(), S.tris.as<rt_triangle>(), nT,
		vertexCount, S.ext.as<float>(), dGeom );
	{
		const uint64_t want = ( vertexCount + TB - 1 ) / TB;
		const int nPart = (int) ( want < 1 ? 1 : want > (uint64_t) BOX_BLOCKS ? (uint64_t) BOX_BLOCKS : want );
		LAUNCH( k_box_partial, dim3( nPart ), dim3( TB ), 0, s, S.verts.as<rt_vertex>(), vertexCount, boxPart );
		LAUNCH( k_box_final, dim3( 1 ), dim3( 64 ), 0, s, S.verts.as<rt_vertex>(), vertexCount, boxPart, nPart, dGeom );
	}

	// The levels.  How many references and nodes a level holds is known on the device only, so the work buffers
	// are sized from the triangle count (and from what earlier builds of this thread needed); a build that
	// outgrows one says so in its state, the buffer is doubled and the levels run again.
	{
		Caps want;
		want.refs = cap_of( 6ull * nT + 65536 );
		want.levelNodes = cap_of( (uint64_t) nT + 4096 );
		want.nodesTotal = cap_of( 4ull * nT + 4096 );
		want.leaf = cap_of( 8ull * nT + 65536 );
		if ( S.caps.refs < want.refs ) S.caps.refs = want.refs;
		if ( S.caps.levelNodes < want.levelNodes ) S.caps.levelNodes = want.levelNodes;
		if ( S.caps.nodesTotal < want.nodesTotal ) S.caps.nodesTotal = want.nodesTotal;
		if ( S.caps.leaf < want.leaf ) S.caps.leaf = want.leaf;
	}
	for ( int attempt = 0; ; ++attempt )
	{
		const Caps caps = S.caps;
		NodeTab tab[2];
		if ( int rc = S.refsA.reserve( 2 * sizeof( uint32_t ) * (size_t) caps.refs, s ) ) return rc;
		if ( int rc = S.refsB.reserve( 2 * sizeof( uint32_t ) * (size_t) caps.refs, s ) ) return rc;
		if ( int rc = node_tab( S.tabA, caps.levelNodes, s, tab[0] ) ) return rc;
		if ( int rc = node_tab( S.tabB, caps.levelNodes, s, tab[1] ) ) return rc;
		if ( int rc = S.info.reserve( sizeof( uint2 ) * (size_t) caps.nodesTotal, s ) ) return rc;
		if ( int rc = S.leafPool.reserve( sizeof( uint32_t ) * ( (size_t) caps.leaf + 1 ), s ) ) return rc;
		{
			const void *before = S.hist.p;
			if ( int rc = S.hist.reserve( sizeof( uint32_t ) * HIST * (size_t) caps.levelNodes, s ) ) return rc;
			if ( S.hist.p != before || !S.histClean )
				GCHK( hipMemsetAsync( S.hist.p, 0, S.hist.cap, s ) );
			S.histClean = false;
		}
		uint32_t *refBuf[2] = { S.refsA.as<uint32_t>(), S.refsB.as<uint32_t>() };
		if ( nT ) LAUNCH( k_iota, dim3( grid_of( nT ) ), dim3( TB ), 0, s, refBuf[0], refBuf[0] + caps.refs, nT );
		GCHK( hipMemsetAsync( &dGeom->lostInPlane, 0, sizeof( uint32_t ), s ) );
		GCHK( hipMemsetAsync( dState, 0, sizeof( BuildState ), s ) );
		LAUNCH( k_root, dim3( 1 ), dim3( 64 ), 0, s, dState, tab[0], dGeom, nT, maxDepth, maxPrims );
		for ( int depth = 0; depth <= maxDepth; ++depth )
		{
			const NodeTab &cur = tab[depth & 1], &next = tab[( depth + 1 ) & 1];
			const uint32_t *refs = refBuf[depth & 1], *refNode = refs + caps.refs;
			uint32_t *nRefs = refBuf[( depth + 1 ) & 1], *nRefNode = nRefs + caps.refs;
			if ( nT && depth == 0 )
				LAUNCH( k_hist, dim3( G_REF ), dim3( TB ), 0, s, dState, depth, cur, S.hist.as<uint32_t>(), refs, refNode, S.ext.as<float>(), nT );
			LAUNCH( k_split, dim3( G_NODE * SPLIT_FINER ), dim3( TB ), 0, s, dState, depth, cur, S.hist.as<uint32_t>(), nodeSums );
			LAUNCH( k_children_count, dim3( G_NODE + ( nT ? G_REF : 0 ) ), dim3( TB ), 0, s, dState, depth, cur, next, nodeSums, S.info.as<uint2>(), caps,
				maxDepth, maxPrims, refs, refNode, S.ext.as<float>(), nT, refSums );
			if ( nT )
				LAUNCH( k_scatter, dim3( G_REF ), dim3( TB ), 0, s, dState, depth, cur, next, refs, refNode, S.ext.as<float>(), nT, refSums,
					nRefs, nRefNode, S.leafPool.as<uint32_t>(), S.hist.as<uint32_t>(), &dGeom->lostInPlane );
		}
		stamp( "levels enqueued" );
		GCHK( hipMemcpyAsync( hState, dState, sizeof( BuildState ), hipMemcpyDeviceToHost, s ) );
		GCHK( hipMemcpyAsync( hGeom, dGeom, sizeof( GeomInfo ), hipMemcpyDeviceToHost, s ) );
		GCHK( hipGetLastError() );
		GCHK( hipStreamSynchronize( s ) );
		stamp( "levels done" );
		S.histClean = true;
		if ( hGeom->badIndex )
			return 3;
		if ( hState->overflow == OVF_NONE )
			break;
		uint32_t *grow = hState->overflow == OVF_REFS ? &S.caps.refs : hState->overflow == OVF_LEVEL_NODES ? &S.caps.levelNodes :
			hState->overflow == OVF_NODES ? &S.caps.nodesTotal : hState->overflow == OVF_LEAF ? &S.caps.leaf : nullptr;
		if ( !grow || *grow >= ( 1u << 30 ) || attempt >= 64 )
			return 4;          // 30-bit links of the 8-byte node
		// (a level that did not fit says how much it holds; the levels below it will hold about as much again)
		const uint64_t twice = 2ull * *grow, asked = 2ull * hState->need;
		*grow = cap_of( twice > asked ? twice : asked );
	}
	out->maxMaterial = hGeom->maxMat;
	out->bbox.center.x = hGeom->box[0]; out->bbox.center.y = hGeom->box[1]; out->bbox.center.z = hGeom->box[2];
	out->bbox.extents.x = hGeom->box[3]; out->bbox.extents.y = hGeom->box[4]; out->bbox.extents.z = hGeom->box[5];
	const int nLevels = (int) hState->nLevels;
	const uint64_t nodesTotal = hState->nodesTotal, leafRefs = hState->leafRefs;
	if ( nLevels < 1 || nLevels > maxDepth + 1 || nodesTotal == 0 )
	{
		snprintf( g_gpu_build_err, sizeof( g_gpu_build_err ), "kd build left an inconsistent state (%d levels, %llu nodes)", nLevels,
			(unsigned long long) nodesTotal );
		return 10;
	}

	// sizes bottom-up, positions top-down
	const uint32_t nNodes = (uint32_t) nodesTotal;
	if ( int rc = S.sub.reserve( 5 * sizeof( uint32_t ) * (size_t) nNodes, s ) ) return rc;
	uint32_t *subNodes = S.sub.as<uint32_t>(), *subRefs = subNodes + nNodes, *subDepth = subRefs + nNodes, *pos = subDepth + nNodes, *refPos = pos + nNodes;
	const uint2 *info = S.info.as<uint2>();
	for ( int d = nLevels - 1; d >= 0; --d )
		LAUNCH( k_sizes, dim3( grid_of( hState->lv[d].nodes ) ), dim3( TB ), 0, s, info, hState->lv[d].nodeBase, hState->lv[d].nodes,
			subNodes, subRefs, subDepth );
	GCHK( hipMemsetAsync( pos, 0, sizeof( uint32_t ), s ) );
	GCHK( hipMemsetAsync( refPos, 0, sizeof( uint32_t ), s ) );
	for ( int d = 0; d < nLevels; ++d )
		LAUNCH( k_positions, dim3( grid_of( hState->lv[d].nodes ) ), dim3( TB ), 0, s, info, hState->lv[d].nodeBase, hState->lv[d].nodes,
			subNodes, subRefs, pos, refPos );
	uint32_t rootDepth = 0;
	GCHK( hipMemcpyAsync( &rootDepth, subDepth, sizeof( uint32_t ), hipMemcpyDeviceToHost, s ) );

	// output buffers (owned by the caller from here on)
	const uint32_t nRecs = (uint32_t) leafRefs;
	// pruned lists: the complete ones are kept behind them (read by rays parallel to an axis, see rt_kdtree.h)
	const bool twoLists = opts->pruneLeaves > 0.0f && nRecs != 0u;
	if ( twoLists && 2ull * nRecs >= ( 1ull << 30 ) )
	{
		snprintf( g_gpu_build_err, sizeof( g_gpu_build_err ), "pruned and complete leaf lists together exceed the 30-bit record index" );
		return 4;
	}
	const size_t nAll = (size_t) nRecs * ( twoLists ? 2 : 1 );
	{
		hipError_t ea = hipMalloc( &out->nodes, sizeof( rt_hip_kdnode ) * (size_t) nNodes );
		if ( ea == hipSuccess ) ea = hipMalloc( &out->recs, sizeof( rt_hip_trirec ) * ( nAll + 1 ) );
		if ( ea == hipSuccess ) ea = hipMalloc( &out->shade, sizeof( rt_hip_trishade ) * ( nT ? nT : 1 ) );
		if ( ea == hipSuccess && twoLists ) ea = hipMalloc( &out->leafFull, sizeof( rt_hip_leaf_full ) * (size_t) nNodes );
		if ( ea != hipSuccess )
		{
			snprintf( g_gpu_build_err, sizeof( g_gpu_build_err ), "allocating the tree failed: %s", hipGetErrorString( ea ) );
			rt_kdtree_device_release( out );
			return 10;
		}
	}
	stamp( "outputs allocated" );
	if ( int rc = S.order.reserve( sizeof( uint32_t ) * ( (size_t) nRecs + 1 ), s ) ) { rt_kdtree_device_release( out ); return rc; }
	LAUNCH_OR( rt_kdtree_device_release( out ), k_emit, dim3( grid_of( nNodes ) ), dim3( TB ), 0, s, info, nNodes, pos, refPos, S.leafPool.as<uint32_t>(),
		(rt_hip_kdnode *) out->nodes, S.order.as<uint32_t>() );
	LAUNCH_OR( rt_kdtree_device_release( out ), k_records, dim3( grid_of( (size_t) nRecs + 1 ) ), dim3( TB ), 0, s, S.verts.as<rt_vertex>(), S.tris.as<rt_triangle>(),
		S.order.as<uint32_t>(), nRecs, (rt_hip_trirec *) out->recs );
	if ( nT )
		LAUNCH_OR( rt_kdtree_device_release( out ), k_shade_records, dim3( grid_of( nT ) ), dim3( TB ), 0, s, S.verts.as<rt_vertex>(), S.tris.as<rt_triangle>(), nT,
			(rt_hip_trishade *) out->shade );
	if ( twoLists )
	{
		rt_hip_trirec *recs = (rt_hip_trirec *) out->recs;
		LAUNCH_OR( rt_kdtree_device_release( out ), k_leaf_full, dim3( grid_of( nNodes ) ), dim3( TB ), 0, s, (const rt_hip_kdnode *) out->nodes, nNodes, nRecs,
			(rt_hip_leaf_full *) out->leafFull );
		GCHK( hipMemcpyAsync( recs + nRecs, recs, sizeof( rt_hip_trirec ) * (size_t) nRecs, hipMemcpyDeviceToDevice, s ) );
		GCHK( hipMemsetAsync( recs + 2 * (size_t) nRecs, 0, sizeof( rt_hip_trirec ), s ) );       // the spare record at the very end
		LAUNCH_OR( rt_kdtree_device_release( out ), k_prune_leaves, dim3( grid_of( nNodes ) ), dim3( TB ), 0, s, (rt_hip_kdnode *) out->nodes, nNodes,
			recs, dGeom, opts->pruneLeaves );
	}
	hipError_t e = hipGetLastError();
	if ( e == hipSuccess ) e = hipStreamSynchronize( s );
	if ( e != hipSuccess )
	{
		snprintf( g_gpu_build_err, sizeof( g_gpu_build_err ), "kd build kernels failed: %s", hipGetErrorString( e ) );
		rt_kdtree_device_release( out );
		return 10;
	}
	stamp( "done" );
	out->nNodes = nNodes;
	out->nRecs = nRecs;
	out->nRecsFull = twoLists ? nRecs : 0;
	out->nTris = nT;
	out->depth = (int) rootDepth;
	return 0;
}

extern "C" int rt_kdtree_device_download( const rt_kdtree_device *t, rt_kdtree_flat *out )
{
	if ( !t || !out || !t->nodes )
		return 1;
	memset( out, 0, sizeof( *out ) );
	out->nNodes = t->nNodes;
	out->nRecs = t->nRecs;
	out->nTris = t->nTris;
	out->bbox = t->bbox;
	out->depth = t->depth;
	const uint64_t nAll = t->nRecs + ( t->leafFull ? t->nRecsFull : 0 );
	out->nRecsFull = nAll - t->nRecs;
	out->nodes = (rt_hip_kdnode *) malloc( sizeof( rt_hip_kdnode ) * ( t->nNodes ? t->nNodes : 1 ) );
	out->recs = (rt_hip_trirec *) malloc( sizeof( rt_hip_trirec ) * ( nAll ? nAll : 1 ) );
	out->shade = (rt_hip_trishade *) malloc( sizeof( rt_hip_trishade ) * ( t->nTris ? t->nTris : 1 ) );
	hipError_t e = hipMemcpy( out->nodes, t->nodes, sizeof( rt_hip_kdnode ) * t->nNodes, hipMemcpyDeviceToHost );
	if ( e == hipSuccess && nAll ) e = hipMemcpy( out->recs, t->recs, sizeof( rt_hip_trirec ) * nAll, hipMemcpyDeviceToHost );
	if ( e == hipSuccess && t->leafFull )
	{
		out->leafFull = (rt_hip_leaf_full *) malloc( sizeof( rt_hip_leaf_full ) * ( t->nNodes ? t->nNodes : 1 ) );
		e = hipMemcpy( out->leafFull, t->leafFull, sizeof( rt_hip_leaf_full ) * t->nNodes, hipMemcpyDeviceToHost );
	}
	if ( e == hipSuccess && t->nTris ) e = hipMemcpy( out->shade, t->shade, sizeof( rt_hip_trishade ) * t->nTris, hipMemcpyDeviceToHost );
	if ( e != hipSuccess )
	{
		snprintf( g_gpu_build_err, sizeof( g_gpu_build_err ), "tree download failed: %s", hipGetErrorString( e ) );
		rt_kdtree_flat_release( out );
		return 10;
	}
	return 0;
}
