// kd-tree builder on the GPU (gfx950): the same tree as the host builder
// (kd_build.cpp, i.e. the reference's rule, rt_funcs_render_pipe.c:591-1209),
// produced level by level in HBM and emitted directly in the layout the
// traversal kernels read, so a rebuilt tree never crosses PCIe.
//
// One level = the set of nodes of one depth; a node is decided (leaf, or split
// axis + 32 candidate planes) when its parent creates it.  Per level, three launches:
//   k_split           node : SAH cost of the 32 planes from the node's histogram, lowest-index minimum - for the
//                            nodes that did not get their split from the kernel that counted them; block sums
//   k_children_count  node : child numbers and reference ranges (scan over the level), child boxes, leaf / inner
//                            decision for the children, the level's records of the tree, the next level's state
//                     ref  : (other blocks of the same launch) left / right counts per block
//   k_scatter         ref  : STABLE partition of the triangle references (a reference goes left if min < sep,
//                            right if max > sep, possibly both); on the way the histogram of every reference
//                            against the planes of the child it goes to, and the split of every child whose
//                            parent's references lie inside one tile, from histograms in LDS
// (k_hist counts the root's references).  Then, over the finished tree: pre-order positions (left child =
// own index + 1) for all nodes at once, nodes + leaf records + shading records.
//
// The level loop runs without the host: level counts live in a state block in device memory, launches have
// fixed grids, work buffers are sized from the triangle count (a level that does not fit says so, the buffer is
// doubled and the levels run again).  The result's buffers are allocated while the levels run, sized by the
// previous build of as many triangles, and filled before the one synchronisation of the build; the first build
// of a size allocates them after it and synchronises once more.
//
// fp32 expressions are the host builder's, operand for operand, and the file
// is compiled with -ffp-contract=off: the result is the host tree bit for bit
// (tests/test_kdtree_gpu.py).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
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
#define LAUNCH( kernel, ... ) do { hipLaunchKernelGGL( kernel, __VA_ARGS__ ); hipError_t e_ = hipGetLastError(); if ( e_ != hipSuccess ) { \
	snprintf( g_gpu_build_err, sizeof( g_gpu_build_err ), "launch of %s failed: %s (%s:%d)", #kernel, hipGetErrorString( e_ ), __FILE__, __LINE__ ); \
	return 10; } } while ( 0 )

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

// Extents of every triangle (one array per axis and bound: references that are neighbours in a node's list are mostly
// neighbours in triangle order, and their loads share cache lines; six floats side by side per triangle made the
// partition twice as slow); the index check and the highest material slot ride
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
	uint32_t nLong;             // leaves of more than 32 references met by k_emit
	uint32_t rootDepth;         // depth of the finished tree
	uint32_t fits;              // the tree is complete and fits the buffers the result is written to (k_fit)
};

struct Caps { uint32_t refs, levelNodes, nodesTotal, leaf; };

// ---- node table of one level (one entry per node of the level) ---------------
enum : uint32_t { K_SPLIT = 1u, K_SORTED = 8u, K_DONE = 16u };     // (bits 1-2: split axis)

struct NodeTab
{
	// What the partition needs of a node, fetched by every reference of the node in one go (written by k_children_count):
	//   [0] over the nodes before this one: x inner nodes, y references handed to children, z references in leaves, w references handed to left children
	//   [1] kind, chosen plane, references going left, first reference
	//   [2] references, kind of the left / right child
	//   [3] first candidate plane and plane distance of the left child, of the right child
	//   [4], [5] what the SAH sweep needs of the left / right child's box (SahBox: c, e, eA, eB)
	uint4 *rec;
	float *box;            // [6 * n]: centre xyz, half extent xyz
	uint32_t *start, *count;
	uint32_t *parent;      // index in the level above
	uint32_t *kind;        // K_SPLIT: inner node; axis; K_SORTED: candidate planes ascend (always, except for NaN / inverted boxes); K_DONE: split chosen by the kernel that counted
	float *mn, *delta;     // candidate plane k = mn + k * delta
	float *sep;            // chosen plane
	uint32_t *cntL, *cntR;
};
constexpr int REC = 6;                   // uint4 per node record
constexpr int NODE_TAB_WORDS = 4 * REC + 6 + 9;

constexpr int HIST = 2 * PARTS + 2;      // per node: left histogram [33], right histogram [33] (sorted planes); else L[32], R[32] at 0 and 33
constexpr int HSLOTS = 64;               // nodes of one tile of references counted in LDS
constexpr int G_REF = 4096, G_NODE = 256;  // blocks of the per-reference / per-node part of a launch
constexpr int RPT = 1, TILE = RPT * TB;  // the partition: references per thread, per tile
constexpr int SPLIT_FINER = 8;           // k_split's blocks take an eighth of a k_children block's nodes each
static_assert( PARTS == 32, "the split search gives one plane to each lane of half a wave" );

// A node's box by value (named members, picked by comparing the axis: an array indexed by the axis lives in scratch memory)
struct Box
{
	float cx, cy, cz, ex, ey, ez;      // centre, half extent
};
__device__ inline Box load_box( const float *b ) { return Box{ b[0], b[1], b[2], b[3], b[4], b[5] }; }
__device__ inline void store_box( float *b, const Box &v ) { b[0] = v.cx; b[1] = v.cy; b[2] = v.cz; b[3] = v.ex; b[4] = v.ey; b[5] = v.ez; }
__device__ inline float pick3( int ax, float x, float y, float z ) { float r = x; if ( ax == 1 ) r = y; if ( ax == 2 ) r = z; return r; }
__device__ inline float box_c( const Box &b, int ax ) { return pick3( ax, b.cx, b.cy, b.cz ); }
__device__ inline float box_e( const Box &b, int ax ) { return pick3( ax, b.ex, b.ey, b.ez ); }

// contiguous share of n nodes for block blk of nBlocks, a multiple of TB long
__device__ inline uint32_t node_share( uint32_t n, uint32_t nBlocks ) { return ( ( n + nBlocks * TB - 1u ) / ( nBlocks * TB ) ) * TB; }

__device__ inline void share_of( uint32_t n, uint32_t per, uint32_t blk, uint32_t &b0, uint32_t &b1 )
{
	const uint64_t s = (uint64_t) blk * per;
	b0 = s < n ? (uint32_t) s : n;
	b1 = s + per < n ? (uint32_t) ( s + per ) : n;
}

// Contiguous share of a level's references for block blk of nBlocks, in tiles of `tile`.  Near the root a node holds so
// many references that its counts would be flushed by thousands of blocks, one atomic after the other on the same 66
// words: there a block takes more tiles (it keeps the counts of a node in LDS from tile to tile), so that a node
// is shared by about 512 blocks at most.
__device__ inline void ref_share( const LevelState &L, uint32_t nBlocks, uint32_t blk, uint32_t tile, uint32_t &b0, uint32_t &b1 )
{
	uint32_t tiles = ( L.refs + nBlocks * tile - 1u ) / ( nBlocks * tile );
	const uint32_t perNode = L.refs / ( L.nodes ? L.nodes : 1u ) / ( 512u * tile );
	if ( tiles < perNode ) tiles = perNode;
	share_of( L.refs, tiles * tile, blk, b0, b1 );
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
struct Decision
{
	uint32_t kind;
	float mn, delta;
};

__device__ __forceinline__ Decision decide_node( const NodeTab &T, uint32_t i, const Box &b, uint32_t cnt, int depth, int maxDepth, int maxPrims )
{
	Decision d = { 0u, 0.0f, 0.0f };
	if ( cnt <= (uint32_t) maxPrims || depth == maxDepth )
	{
		T.kind[i] = 0u;
		return d;
	}
	float range = b.ex;
	int ax = 0;
	if ( range < b.ey ) { range = b.ey; ax = 1; }
	if ( range < b.ez ) { range = b.ez; ax = 2; }
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
	d.kind = K_SPLIT | ( (uint32_t) ax << 1 ) | ( sorted ? K_SORTED : 0u );
	d.mn = mn;
	d.delta = delta;
	T.kind[i] = d.kind;
	T.mn[i] = mn;
	T.delta[i] = delta;
	return d;
}

// the root and the state of level 0 (the state has been zeroed: a level that was never reached holds nothing)
__global__ __launch_bounds__( 64 ) void k_root( BuildState *st, NodeTab T, const GeomInfo *geom, uint32_t nT, int maxDepth, int maxPrims )
{
	if ( threadIdx.x != 0 )
		return;
	const Box b = load_box( geom->box );
	store_box( T.box, b );
	T.start[0] = 0u;
	T.count[0] = nT;
	T.parent[0] = 0u;
	decide_node( T, 0u, b, nT, 0, maxDepth, maxPrims );
	st->lv[0].nodes = 1u;
	st->lv[0].refs = nT;
}

// what the SAH sweep needs to know of a node's box
struct SahBox
{
	float c, e;            // centre and half extent along the split axis
	float eA, eB;          // half extents along the other two
	float mn, delta;
};

__device__ __forceinline__ SahBox sah_box( const Box &b, uint32_t kd, float mn, float delta )
{
	const int ax = (int) ( ( kd >> 1 ) & 3u );
	SahBox s;
	s.c = box_c( b, ax );
	s.e = box_e( b, ax );
	s.eA = pick3( ax, b.ey, b.ex, b.ex );
	s.eB = pick3( ax, b.ez, b.ez, b.ey );
	s.mn = mn;
	s.delta = delta;
	return s;
}

__device__ inline SahBox sah_box( const NodeTab &T, uint32_t i, uint32_t kd )
{
	return sah_box( load_box( T.box + 6ull * i ), kd, T.mn[i], T.delta[i] );
}

// SAH sweep of :591-857 on the counts of one node, by 32 lanes (lane l = plane l):
// the lowest index with the strictly smallest cost.  h: the node's histogram (LDS or HBM).
__device__ inline void sah_node( const NodeTab &T, uint32_t i, uint32_t kd, const SahBox &B, const uint32_t *h, uint32_t l )
{
	const bool sorted = ( kd & K_SORTED ) != 0u;
	const float c = B.c, e = B.e, mn = B.mn, delta = B.delta, eA = B.eA, eB = B.eB;
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

// One triangle's extent [a, b] along the axis of node i against the node's 32 planes mn + k * delta: into the node's slot of the LDS histograms
// when it has one (slot = node index - first node of the tile), else into its histogram in HBM.
struct Ext
{
	const float *p;
	uint32_t nT;
	__device__ float lo( uint32_t ax, uint32_t tri ) const { return p[(size_t) ax * nT + tri]; }
	__device__ float hi( uint32_t ax, uint32_t tri ) const { return p[( 3ull + ax ) * nT + tri]; }
};

__device__ inline void hist_add( uint32_t i, uint32_t kd, float mn, float delta, float a, float b, uint32_t first, uint32_t *sh, uint32_t *hist )
{
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

// what finish_slots knows of the node in a slot
struct SlotFacts
{
	uint32_t kind;         // 0: nothing to do; K_DONE set here: all its references were in this tile
	SahBox box;
};

// One thread per slot: what finish_slots needs of the nodes first .. first + nSlots - 1 of table T
// (complete( i ): all references of node i were in this tile).
template <class Complete>
__device__ inline void gather_facts( const NodeTab &T, uint32_t first, uint32_t nSlots, SlotFacts *facts, Complete complete )
{
	if ( threadIdx.x < nSlots )
	{
		const uint32_t i = first + threadIdx.x, kd = T.kind[i];
		uint32_t fk = 0u;
		if ( kd & K_SPLIT )
		{
			fk = kd;
			if ( complete( i ) )
			{
				fk = kd | K_DONE;
				facts[threadIdx.x].box = sah_box( T, i, kd );
				T.kind[i] = fk;
			}
		}
		facts[threadIdx.x].kind = fk;
	}
}

// After a tile of references has been counted and the block has met: the nodes that have a slot in LDS (nodes
// first .. first + nSlots - 1 of table T, facts[] filled in).  A node all of whose references were in this tile gets
// its split chosen right here, by half a wave, and never touches the histograms in HBM; the first `keep` slots stay
// as they are (their nodes go on in the next tile); the counts of the others are added to theirs in HBM.  The facts
// are left cleared.
__device__ inline void finish_slots( const NodeTab &T, uint32_t first, uint32_t nSlots, uint32_t keep, uint32_t *sh, uint32_t *hist, SlotFacts *facts )
{
	const uint32_t l = threadIdx.x & 31u;
	for ( uint32_t s = threadIdx.x >> 5; s < nSlots; s += TB / 32 )
	{
		const uint32_t i = first + s, kd = facts[s].kind;
		if ( !( kd & K_SPLIT ) )
			continue;
		uint32_t *h = sh + s * HIST;
		if ( kd & K_DONE )
		{
			sah_node( T, i, kd, facts[s].box, h, l );
			hist_clear( h, l );
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
		if ( l == 0u )
			facts[s].kind = 0u;
	}
}

// Histogram of every triangle extent against the 32 planes of its node, for the root level (below it the
// partition counts for the level it fills, k_scatter).  References are ordered by node, a block takes a contiguous
// run of them in tiles of TB; the nodes a tile touches are counted in LDS (the first HSLOTS of them).
__global__ __launch_bounds__( TB ) void k_hist( const BuildState *st, int depth, NodeTab T, uint32_t *hist, const uint32_t *refs,
	const uint32_t *refNode, Ext ext )
{
	if ( st->overflow != OVF_NONE )
		return;
	const LevelState L = st->lv[depth];
	uint32_t b0, b1;
	ref_share( L, gridDim.x, blockIdx.x, TB, b0, b1 );
	if ( b0 >= b1 )
		return;
	__shared__ uint32_t sh[HSLOTS * HIST];
	__shared__ SlotFacts facts[HSLOTS];
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
			{
				const uint32_t tri = refs[j], ax = ( kd >> 1 ) & 3u;
				hist_add( i, kd, T.mn[i], T.delta[i], ext.lo( ax, tri ), ext.hi( ax, tri ), node0, sh, hist );
			}
		}
		__syncthreads();
		const uint32_t keep = tileEnd < b1 && refNode[tileEnd] == node0 ? 1u : 0u;
		const uint32_t span = nodeLast - node0 + 1u;
		const uint32_t nSlots = span < (uint32_t) HSLOTS ? span : (uint32_t) HSLOTS;
		gather_facts( T, node0, nSlots, facts, [&]( uint32_t i ) { const uint32_t first = T.start[i]; return first >= j0 && first + T.count[i] <= tileEnd; } );
		__syncthreads();
		finish_slots( T, node0, nSlots, keep, sh, hist, facts );
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
			const uint32_t kdn = T.kind[ni];
			sah_node( T, ni, kdn, sah_box( T, ni, kdn ), h, l );
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

// what k_children_count reads of a node
struct NodeFacts
{
	uint32_t kd, cnt, cl, cr, start, parent;
	float sep;
	Box b;
};

__device__ __forceinline__ NodeFacts node_facts( const NodeTab &T, uint32_t i )
{
	NodeFacts f;
	f.kd = T.kind[i];
	f.cnt = T.count[i];
	f.start = T.start[i];
	f.parent = T.parent[i];
	f.cl = T.cntL[i];
	f.cr = T.cntR[i];
	f.sep = T.sep[i];
	f.b = load_box( T.box + 6ull * i );
	return f;
}

// Two independent pieces of work in one launch (both need nothing but k_split's results).
// Blocks [0, G_NODE): number the children (a scan over the level's nodes, in two steps: the sums of k_split's blocks,
// then inside the block), write the child boxes - slab split at sep (:1034-1073) -, decide leaf / inner node for
// them, write this level's records of the tree in level order, and the next level's state.  Every block works out the
// level's totals for itself, so all of them agree whether the next level fits the work buffers.
// Blocks [G_NODE, G_NODE + G_REF): references going left / right (min < sep / max > sep, :971-1029) in every block's
// share of the level (the first step of the partition's scan).
__global__ __launch_bounds__( TB ) void k_children_count( BuildState *st, int depth, NodeTab T, NodeTab N, const uint4 *blockSums, uint2 *info, uint4 *walk,
	Caps caps, int maxDepth, int maxPrims, const uint32_t *refs, const uint32_t *refNode, Ext ext, uint2 *refSums )
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
		ref_share( L, G_REF, blockIdx.x - G_NODE, TILE, b0, b1 );
		uint32_t acc[2] = { 0u, 0u };
		// (four references per thread under way at once: the three rounds of loads of one are all there is to wait for)
		for ( uint32_t j = b0 + threadIdx.x; j < b1; j += 4 * TB )
		{
			uint32_t node[4], tri[4], kd[4];
			float sep[4], lo[4], hi[4];
#pragma unroll
			for ( int u = 0; u < 4; ++u )
			{
				const uint32_t ju = j + u * TB < b1 ? j + u * TB : j;      // (past the end: the first one again, not counted)
				node[u] = refNode[ju];
				tri[u] = refs[ju];
			}
#pragma unroll
			for ( int u = 0; u < 4; ++u )
			{
				kd[u] = T.kind[node[u]];
				sep[u] = T.sep[node[u]];
			}
#pragma unroll
			for ( int u = 0; u < 4; ++u )
			{
				const uint32_t ax = ( kd[u] >> 1 ) & 3u;
				lo[u] = ext.lo( ax, tri[u] );
				hi[u] = ext.hi( ax, tri[u] );
			}
#pragma unroll
			for ( int u = 0; u < 4; ++u )
				if ( ( kd[u] & K_SPLIT ) && j + u * TB < b1 )
				{
					acc[0] += lo[u] < sep[u] ? 1u : 0u;
					acc[1] += hi[u] > sep[u] ? 1u : 0u;
				}
		}
		block_sum( acc, red );
		if ( threadIdx.x == 0 )
			refSums[blockIdx.x - G_NODE] = make_uint2( acc[0], acc[1] );
		return;
	}
	uint32_t b0, b1;
	share_of( n, node_share( n, G_NODE ), blockIdx.x, b0, b1 );
	// what this thread needs of its node in the block's first tile (the loads are in flight while the sums are formed)
	NodeFacts f0 = { 0u, 0u, 0u, 0u, 0u, 0u, 0.0f, { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f } };
	if ( b0 + threadIdx.x < b1 )
		f0 = node_facts( T, b0 + threadIdx.x );
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
	uint32_t run[4] = { pt[0], pt[1], pt[2], pt[3] };
	const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
	const uint32_t nextBase = L.nodeBase + n, parentBase = depth ? st->lv[depth - 1].nodeBase : 0u;
	for ( uint32_t i0 = b0; i0 < b1; i0 += TB )
	{
		const uint32_t i = i0 + threadIdx.x;
		const bool valid = i < b1;
		uint32_t kd = 0u, own[4] = { 0u, 0u, 0u, 0u }, cnt = 0u, cl = 0u, cr = 0u;
		if ( valid )
		{
			if ( i0 != b0 )
				f0 = node_facts( T, i );
			kd = f0.kd;
			cnt = f0.cnt;
			if ( kd & K_SPLIT )
			{
				cl = f0.cl;
				cr = f0.cr;
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
		uint4 *rec = T.rec + (size_t) REC * i;
		rec[0] = make_uint4( ex[0], ex[1], ex[2], ex[3] );
		walk[L.nodeBase + i] = make_uint4( parentBase + f0.parent, ex[0], ex[2], (uint32_t) depth );
		if ( !( kd & K_SPLIT ) )
		{
			rec[1] = make_uint4( 0u, 0u, 0u, f0.start );
			rec[2] = make_uint4( cnt, 0u, 0u, 0u );
			info[L.nodeBase + i] = make_uint2( L.leafBase + ex[2], ( cnt << 2 ) | 3u );
			continue;
		}
		const Box b = f0.b;
		Box bl = b, br = b;
		const int ax = (int) ( ( kd >> 1 ) & 3u );
		const float sep = f0.sep;
		const uint32_t c = 2u * ex[0];
		{
			const float bc = box_c( b, ax ), be = box_e( b, ax );
			const float lc = 0.5f * ( bc - be + sep ), le = 0.5f * ( sep - bc + be );
			const float rc = 0.5f * ( bc + be + sep ), re = 0.5f * ( bc + be - sep );
			if ( ax == 0 ) { bl.cx = lc; bl.ex = le; br.cx = rc; br.ex = re; }
			else if ( ax == 1 ) { bl.cy = lc; bl.ey = le; br.cy = rc; br.ey = re; }
			else { bl.cz = lc; bl.ez = le; br.cz = rc; br.ez = re; }
		}
		store_box( N.box + 6ull * c, bl );
		store_box( N.box + 6ull * c + 6, br );
		N.start[c] = ex[1];
		N.count[c] = cl;
		N.start[c + 1] = ex[1] + cl;
		N.count[c + 1] = cr;
		N.parent[c] = N.parent[c + 1] = i;
		const Decision dl = decide_node( N, c, bl, cl, depth + 1, maxDepth, maxPrims );
		const Decision dr = decide_node( N, c + 1u, br, cr, depth + 1, maxDepth, maxPrims );
		rec[1] = make_uint4( kd, __float_as_uint( sep ), cl, f0.start );
		rec[2] = make_uint4( cnt, dl.kind, dr.kind, 0u );
		rec[3] = make_uint4( __float_as_uint( dl.mn ), __float_as_uint( dl.delta ), __float_as_uint( dr.mn ), __float_as_uint( dr.delta ) );
		if ( dl.kind & K_SPLIT )
		{
			const SahBox s = sah_box( bl, dl.kind, dl.mn, dl.delta );
			rec[4] = make_uint4( __float_as_uint( s.c ), __float_as_uint( s.e ), __float_as_uint( s.eA ), __float_as_uint( s.eB ) );
		}
		if ( dr.kind & K_SPLIT )
		{
			const SahBox s = sah_box( br, dr.kind, dr.mn, dr.delta );
			rec[5] = make_uint4( __float_as_uint( s.c ), __float_as_uint( s.e ), __float_as_uint( s.eA ), __float_as_uint( s.eB ) );
		}
		info[L.nodeBase + i] = make_uint2( __float_as_uint( sep ), ( ( nextBase + c ) << 2 ) | (uint32_t) ax );
	}
}

// Stable partition (:971-1029): the k-th reference of a node that goes left becomes the k-th reference of its left
// child; references of leaves move to the leaf pool.  A reference's rank among those of its node that go left is its
// rank among all of the level that go left, less the left counts of the nodes before its own (NodeTab::rec[0]).
// A reference that goes to a child is counted against that child's planes on the way (the children of the nodes of a
// tile are neighbours in the next level's table N: the same LDS histograms and in-tile splits as in k_hist, a child
// being complete when its parent's references all lie in the tile), so the next level starts with its counts taken.
// The kernel waits for memory most of the time, so a tile is three rounds of loads and no more (reference -> the
// node's record, which holds all that is needed of the node and of its children -> extents), and a thread has RPT
// references of the tile under way at once.
__global__ __launch_bounds__( TB ) void k_scatter( const BuildState *st, int depth, NodeTab T, NodeTab N, const uint32_t *refs, const uint32_t *refNode,
	Ext ext, const uint2 *blockSums, uint32_t *nextRefs, uint32_t *nextRefNode, uint32_t *leafPool,
	uint32_t *hist, uint32_t *lostInPlane )
{
	if ( st->overflow != OVF_NONE )
		return;
	const LevelState L = st->lv[depth];
	uint32_t b0, b1;
	ref_share( L, gridDim.x, blockIdx.x, TILE, b0, b1 );
	if ( b0 >= b1 )
		return;
	__shared__ uint32_t red[( TB / 64 ) * 2];
	__shared__ uint32_t waveTot[( TB / 64 ) * RPT * 2];
	__shared__ uint32_t sh[HSLOTS * HIST];
	__shared__ SlotFacts facts[HSLOTS];
	for ( int k = threadIdx.x; k < HSLOTS * HIST; k += TB )
		sh[k] = 0u;
	if ( threadIdx.x < HSLOTS )
		facts[threadIdx.x].kind = 0u;
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
	for ( uint32_t j0 = b0; j0 < b1; j0 += TILE )
	{
		const uint32_t tileEnd = j0 + TILE < b1 ? j0 + TILE : b1;
		const uint32_t node0 = refNode[j0], nodeLast = refNode[tileEnd - 1u];
		const uint32_t nodeNext = tileEnd < b1 ? refNode[tileEnd] : 0xffffffffu;
		const uint32_t kd0 = T.rec[(size_t) REC * node0 + 1].x, kdLast = T.rec[(size_t) REC * nodeLast + 1].x;
		const uint32_t c0 = 2u * T.rec[(size_t) REC * node0].x, cEnd = 2u * ( T.rec[(size_t) REC * nodeLast].x + ( kdLast & K_SPLIT ) );
		// (the first node of the tile goes on in the next one: its two children keep their counts in LDS)
		const uint32_t keep = nodeNext == node0 && ( kd0 & K_SPLIT ) ? 2u : 0u;
		uint32_t tri[RPT], kind[RPT], cl[RPT], first[RPT];
		uint4 ex[RPT];
		bool left[RPT], right[RPT];
		unsigned long long mL[RPT], mR[RPT];
		// round 1: the references
		uint32_t node[RPT];
		bool speaks[RPT];          // the node's first reference in the tile speaks for the node: what finish_slots needs of its children
		for ( int u = 0; u < RPT; ++u )
		{
			const uint32_t j = j0 + u * TB + threadIdx.x;
			node[u] = tri[u] = 0u;
			speaks[u] = false;
			if ( j < tileEnd )
			{
				node[u] = refNode[j];
				tri[u] = refs[j];
				speaks[u] = j == j0 || refNode[j - 1u] != node[u];
			}
		}
		// round 2: their nodes' records
		uint4 r1[RPT], r2[RPT], r3[RPT];
		for ( int u = 0; u < RPT; ++u )
		{
			const uint4 *rec = T.rec + (size_t) REC * node[u];
			ex[u] = rec[0];
			r1[u] = rec[1];
			r2[u] = rec[2];
			r3[u] = rec[3];
		}
		// round 3: the triangles' extents; the children's boxes where the children get their split right here
		float pa[RPT], pb[RPT], la[RPT], lb[RPT], ra[RPT], rb[RPT];     // along the axis of the node, of its left, of its right child
		uint4 bx[RPT][2];
		bool complete[RPT];
		for ( int u = 0; u < RPT; ++u )
		{
			const uint32_t j = j0 + u * TB + threadIdx.x;
			kind[u] = j < tileEnd ? r1[u].x : 0u;
			cl[u] = r1[u].z;
			first[u] = r1[u].w;
			pa[u] = pb[u] = la[u] = lb[u] = ra[u] = rb[u] = 0.0f;
			bx[u][0] = bx[u][1] = make_uint4( 0u, 0u, 0u, 0u );
			complete[u] = false;
			if ( kind[u] & K_SPLIT )
			{
				const uint32_t ax = ( kind[u] >> 1 ) & 3u, axl = ( r2[u].y >> 1 ) & 3u, axr = ( r2[u].z >> 1 ) & 3u;
				pa[u] = ext.lo( ax, tri[u] );
				pb[u] = ext.hi( ax, tri[u] );
				if ( r2[u].y & K_SPLIT ) { la[u] = ext.lo( axl, tri[u] ); lb[u] = ext.hi( axl, tri[u] ); }
				if ( r2[u].z & K_SPLIT ) { ra[u] = ext.lo( axr, tri[u] ); rb[u] = ext.hi( axr, tri[u] ); }
				complete[u] = speaks[u] && first[u] >= j0 && first[u] + r2[u].x <= tileEnd;
				if ( complete[u] )
				{
					bx[u][0] = T.rec[(size_t) REC * node[u] + 4];
					bx[u][1] = T.rec[(size_t) REC * node[u] + 5];
				}
			}
		}
		for ( int u = 0; u < RPT; ++u )
		{
			left[u] = right[u] = false;
			if ( kind[u] & K_SPLIT )
			{
				const uint32_t c = 2u * ex[u].x;
				const float sep = __uint_as_float( r1[u].y );
				left[u] = pa[u] < sep;
				right[u] = pb[u] > sep;
				if ( left[u] && ( r2[u].y & K_SPLIT ) )
					hist_add( c, r2[u].y, __uint_as_float( r3[u].x ), __uint_as_float( r3[u].y ), la[u], lb[u], c0, sh, hist );
				if ( right[u] && ( r2[u].z & K_SPLIT ) )
					hist_add( c + 1u, r2[u].z, __uint_as_float( r3[u].z ), __uint_as_float( r3[u].w ), ra[u], rb[u], c0, sh, hist );
				if ( speaks[u] )
					for ( uint32_t side = 0u; side < 2u; ++side )
					{
						const uint32_t cn = c + side, slot = cn - c0, kdc = side ? r2[u].z : r2[u].y;
						if ( slot >= (uint32_t) HSLOTS || !( kdc & K_SPLIT ) )
							continue;
						SlotFacts f;
						f.kind = kdc;
						if ( complete[u] )
						{
							const uint4 q = bx[u][side];
							f.kind = kdc | K_DONE;
							f.box.c = __uint_as_float( q.x ); f.box.e = __uint_as_float( q.y );
							f.box.eA = __uint_as_float( q.z ); f.box.eB = __uint_as_float( q.w );
							f.box.mn = __uint_as_float( side ? r3[u].z : r3[u].x );
							f.box.delta = __uint_as_float( side ? r3[u].w : r3[u].y );
							N.kind[cn] = kdc | K_DONE;
						}
						facts[slot] = f;
					}
			}
			mL[u] = __ballot( left[u] );
			mR[u] = __ballot( right[u] );
			if ( lane == 0u )
			{
				waveTot[( u * ( TB / 64 ) + w ) * 2] = (uint32_t) __popcll( mL[u] );
				waveTot[( u * ( TB / 64 ) + w ) * 2 + 1] = (uint32_t) __popcll( mR[u] );
			}
		}
		__syncthreads();
		for ( int u = 0; u < RPT; ++u )
		{
			const uint32_t j = j0 + u * TB + threadIdx.x;
			uint32_t gl = run[0], gr = run[1];
			for ( uint32_t q = 0; q < TB / 64; ++q )
			{
				const uint32_t tl = waveTot[( u * ( TB / 64 ) + q ) * 2], tr = waveTot[( u * ( TB / 64 ) + q ) * 2 + 1];
				if ( q < w ) { gl += tl; gr += tr; }
				run[0] += tl;
				run[1] += tr;
			}
			gl += (uint32_t) __popcll( mL[u] & below );
			gr += (uint32_t) __popcll( mR[u] & below );
			if ( j >= tileEnd )
				continue;
			if ( !( kind[u] & K_SPLIT ) )
				leafPool[L.leafBase + ex[u].z + ( j - first[u] )] = tri[u];
			else
			{
				const uint32_t c = 2u * ex[u].x;
				if ( !left[u] && !right[u] )
					atomicAdd( lostInPlane, 1u );             // (rare: a triangle flat in the split axis that lies in the plane)
				if ( left[u] )
				{
					const uint32_t p = ex[u].y + ( gl - ex[u].w );
					nextRefs[p] = tri[u];
					nextRefNode[p] = c;
				}
				if ( right[u] )
				{
					const uint32_t p = ex[u].y + cl[u] + ( gr - ( ex[u].y - ex[u].w ) );
					nextRefs[p] = tri[u];
					nextRefNode[p] = c + 1u;
				}
			}
		}
		const uint32_t span = cEnd - c0;
		finish_slots( N, c0, span < (uint32_t) HSLOTS ? span : (uint32_t) HSLOTS, keep, sh, hist, facts );
		__syncthreads();
	}
}

// ---- finished tree: sizes, positions, output --------------------------------
// These kernels too take their counts from the state in device memory and are enqueued behind the levels without
// the host knowing how the build went: fixed grids, grid-stride loops, nothing done for a level that does not exist.
constexpr int G_FIN = 1024;

struct SubTree             // per node, in level order
{
	// x: the parent (level order), y: inner nodes before this one in its level, z: references in leaves before this one in
	// its level, w: the level (written by k_children_count)
	uint4 *walk;
	uint32_t *pos, *refPos;             // pre-order number, first leaf reference in pre-order
	uint32_t *longLeaves;               // the leaves k_emit leaves to k_emit_long
};

// Pre-order numbering, left child right behind its parent (:860-912), for all nodes at once.  Level order keeps the
// children of neighbours next to each other, so "everything left of node x" is a prefix [0, r) of every level: the
// nodes left of x's ancestors on the levels above, and below x's level the children of the prefix above - twice
// the inner nodes in it.  x comes after all of those and after its ancestors, and after nothing else.  The same
// walk over the leaf references left of x gives the place of a leaf's list.
__global__ __launch_bounds__( TB ) void k_walk( const BuildState *st, SubTree sub )
{
	if ( st->overflow != OVF_NONE || st->nLevels == 0u )
		return;
	const uint32_t n = st->nodesTotal, nL = st->nLevels;
	for ( uint32_t id = blockIdx.x * TB + threadIdx.x; id < n; id += gridDim.x * TB )
	{
		const uint4 me = sub.walk[id];
		const uint32_t d = me.w;
		uint32_t p = d, rp = 0u;
		uint4 w = me;
		for ( uint32_t dd = d, cur = id; ; --dd )
		{
			p += cur - st->lv[dd].nodeBase;
			rp += w.z;
			if ( dd == 0u )
				break;
			cur = w.x;
			w = sub.walk[cur];
		}
		uint32_t inner = me.y;
		for ( uint32_t dd = d + 1u; dd < nL && inner != 0u; ++dd )
		{
			const uint32_t r = 2u * inner;
			p += r;
			if ( r < st->lv[dd].nodes )
			{
				const uint4 q = sub.walk[st->lv[dd].nodeBase + r];
				rp += q.z;
				inner = q.y;
			}
			else
			{
				rp += st->lv[dd + 1u].leafBase - st->lv[dd].leafBase;
				inner = st->lv[dd + 1u].nodes / 2u;
			}
		}
		sub.pos[id] = p;
		sub.refPos[id] = rp;
	}
}

// Is there a finished tree, and does it fit the buffers the kernels below write to?  (They are allocated while the
// levels run, from what the previous build came to; capNodes = 0: not yet.)
__global__ void k_fit( BuildState *st, const GeomInfo *geom, uint32_t capNodes, uint32_t capRecs, uint32_t twoLists )
{
	const bool built = st->overflow == OVF_NONE && st->nLevels != 0u && geom->badIndex == 0u;
	st->rootDepth = built ? st->nLevels - 1u : 0u;
	st->nLong = 0u;
	st->fits = built && st->nodesTotal <= capNodes && st->leafRefs <= capRecs && ( !twoLists || 2ull * st->leafRefs < ( 1ull << 30 ) ) ? 1u : 0u;
}

__global__ __launch_bounds__( TB ) void k_emit( BuildState *st, const uint2 *info, SubTree sub, const uint32_t *leafPool, rt_hip_kdnode *nodes,
	uint32_t *order )
{
	if ( !st->fits )
		return;
	const uint32_t n = st->nodesTotal;
	for ( uint32_t id = blockIdx.x * TB + threadIdx.x; id < n; id += gridDim.x * TB )
	{
		const uint2 v = info[id];
		rt_hip_kdnode o;
		if ( ( v.y & 3u ) == 3u )
		{
			const uint32_t cnt = v.y >> 2, dst = sub.refPos[id];
			o.a = dst;
			o.b = v.y;
			if ( cnt <= 32u )
				for ( uint32_t k = 0; k < cnt; ++k )
					order[dst + k] = leafPool[v.x + k];
			else
				sub.longLeaves[atomicAdd( &st->nLong, 1u )] = id;
		}
		else
		{
			o.a = v.x;
			o.b = ( sub.pos[( v.y >> 2 ) + 1u] << 2 ) | ( v.y & 3u );
		}
		nodes[sub.pos[id]] = o;
	}
}

// the leaf lists of more than 32 references (leaves at the depth limit), a wave per leaf
__global__ __launch_bounds__( TB ) void k_emit_long( const BuildState *st, const uint2 *info, SubTree sub, const uint32_t *leafPool, uint32_t *order )
{
	if ( !st->fits )
		return;
	const uint32_t n = st->nLong, lane = threadIdx.x & 63u, waves = gridDim.x * ( TB / 64 );
	for ( uint32_t q = blockIdx.x * ( TB / 64 ) + ( threadIdx.x >> 6 ); q < n; q += waves )
	{
		const uint32_t id = sub.longLeaves[q];
		const uint2 v = info[id];
		const uint32_t cnt = v.y >> 2, dst = sub.refPos[id];
		for ( uint32_t k = lane; k < cnt; k += 64u )
			order[dst + k] = leafPool[v.x + k];
	}
}

__global__ __launch_bounds__( TB ) void k_records( const BuildState *st, const rt_vertex *verts, const rt_triangle *tris, const uint32_t *order,
	rt_hip_trirec *recs )
{
	if ( !st->fits )
		return;
	const uint32_t nRecs = st->leafRefs;
	for ( uint32_t i = blockIdx.x * TB + threadIdx.x; i <= nRecs; i += gridDim.x * TB )
	{
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
}

// (needs no tree, only indices that have been checked: k_extents ran before)
__global__ __launch_bounds__( TB ) void k_shade_records( const GeomInfo *geom, const rt_vertex *verts, const rt_triangle *tris, uint32_t nT,
	rt_hip_trishade *shade )
{
	const uint32_t i = blockIdx.x * TB + threadIdx.x;
	if ( i >= nT || geom->badIndex != 0u )
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
__global__ __launch_bounds__( TB ) void k_prune_leaves( const BuildState *st, rt_hip_kdnode *nodes, rt_hip_trirec *recs, const GeomInfo *geom, float margin )
{
	const uint32_t leaf = blockIdx.x * TB + threadIdx.x;
	if ( !st->fits || leaf >= st->nodesTotal || geom->lostInPlane != 0u )
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
__global__ __launch_bounds__( TB ) void k_leaf_full( const BuildState *st, const rt_hip_kdnode *nodes, rt_hip_leaf_full *full )
{
	const uint32_t i = blockIdx.x * TB + threadIdx.x;
	if ( !st->fits || i >= st->nodesTotal )
		return;
	const uint32_t shift = st->leafRefs;
	const rt_hip_kdnode n = nodes[i];
	rt_hip_leaf_full f = { 0u, 0u };
	if ( ( n.b & 3u ) == 3u )
	{
		f.first = n.a + shift;
		f.count = n.b >> 2;
	}
	full[i] = f;
}

// the complete record array once more behind itself (16 bytes per thread and step), and the spare record at the very end
__global__ __launch_bounds__( TB ) void k_copy_full( const BuildState *st, rt_hip_trirec *recs )
{
	static_assert( sizeof( rt_hip_trirec ) % sizeof( uint4 ) == 0, "records are copied in 16-byte pieces" );
	if ( !st->fits )
		return;
	const size_t per = sizeof( rt_hip_trirec ) / sizeof( uint4 ), n = (size_t) st->leafRefs * per;
	const uint4 *from = (const uint4 *) recs;
	uint4 *to = (uint4 *) ( recs + st->leafRefs );
	for ( size_t i = (size_t) blockIdx.x * TB + threadIdx.x; i < n + per; i += (size_t) gridDim.x * TB )
		to[i] = i < n ? from[i] : make_uint4( 0u, 0u, 0u, 0u );
}

inline unsigned grid_of( size_t n ) { return (unsigned) ( ( n + TB - 1 ) / TB ); }

struct Scratch
{
	DVec verts, tris, ext, refsA, refsB, tabA, tabB, hist, sums, info, leafPool, sub, order, geom, state;
	Caps caps = { 0u, 0u, 0u, 0u };
	uint32_t lastNT = 0xffffffffu, lastNodes = 0, lastRecs = 0;      // what the previous build of this thread came to
	// buffers of a tree that is no longer in use, kept for the next result (rt_kdtree_device_recycle): [0] nodes, [1] records,
	// [2] shading records, [3] complete-list table, with their sizes as allocated
	void *spare[4] = { nullptr, nullptr, nullptr, nullptr };
	size_t spareBytes[4] = { 0, 0, 0, 0 };
	void drop_spare()
	{
		for ( int k = 0; k < 4; ++k )
		{
			if ( spare[k] ) (void) hipFree( spare[k] );
			spare[k] = nullptr;
			spareBytes[k] = 0;
		}
	}
	// `bytes` of device memory: the spare buffer of that kind if it is large enough (and not wastefully so), else new
	hipError_t take( int kind, void **p, size_t bytes )
	{
		if ( spare[kind] && spareBytes[kind] >= bytes && spareBytes[kind] <= 2 * bytes + ( 1u << 20 ) )
		{
			*p = spare[kind];
			spare[kind] = nullptr;
			spareBytes[kind] = 0;
			return hipSuccess;
		}
		return hipMalloc( p, bytes );
	}
	bool histClean = false;      // every entry of `hist` is zero (each level leaves it so; a build that failed half-way may not)
	void *pinned = nullptr;      // BuildState + GeomInfo as the host reads them
	// geometry copied ahead of its build (rt_kdtree_device_stage): the ticket that names it, how many leading entries
	// of verts / tris hold it, and an event behind the last of its copies for a build or staging on another stream
	uint64_t stageTicket = 0, stagedV = 0, stagedT = 0;
	hipStream_t stageStream = nullptr;
	hipEvent_t stageDone = nullptr;
	bool stagePending = false;
	void forget_staged() { stageTicket = 0; stagedV = stagedT = 0; }
	void release()
	{
		forget_staged();
		stagePending = false;
		if ( stageDone ) (void) hipEventDestroy( stageDone );      // (hipFree below waits for the copies themselves)
		stageDone = nullptr;
		for ( DVec *v : { &verts, &tris, &ext, &refsA, &refsB, &tabA, &tabB, &hist, &sums, &info, &leafPool, &sub, &order, &geom, &state } )
			v->release();
		drop_spare();
		if ( pinned ) (void) hipHostFree( pinned );
		pinned = nullptr;
		caps = Caps{ 0u, 0u, 0u, 0u };
		histClean = false;
		lastNT = 0xffffffffu;
	}
};
thread_local Scratch g_scratch;
thread_local int g_scratchDev = -1;

// carve a level's node table out of one allocation
int node_tab( DVec &buf, size_t n, hipStream_t s, NodeTab &t )
{
	if ( int rc = buf.reserve( n * NODE_TAB_WORDS * sizeof( uint32_t ), s ) ) return rc;
	uint32_t *p = buf.as<uint32_t>();
	t.rec = (uint4 *) p; p += 4 * REC * n;          // first: 16-byte aligned
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

// the scratch of this thread follows the current device
static int scratch_on_current_device()
{
	int dev = 0;
	GCHK( hipGetDevice( &dev ) );
	if ( g_scratchDev != dev )
	{
		g_scratch.release();
		g_scratchDev = dev;
	}
	return 0;
}

// copies staged on another stream come first: they write the buffers this stream is about to write
static int wait_for_staged( Scratch &S, hipStream_t s )
{
	if ( S.stagePending && S.stageStream != s )
		GCHK( hipStreamWaitEvent( s, S.stageDone, 0 ) );
	return 0;
}

extern "C" int rt_kdtree_device_stage( uint64_t *ticket, const rt_vertex *verts, uint64_t nVerts, const rt_triangle *tris,
	uint64_t nTris, void *hipStream )
{
	static std::atomic<uint64_t> lastTicket{ 0 };      // (one sequence for all threads: a ticket shown to another thread's builder must not match one of that thread's own)
	if ( !ticket || ( nVerts && !verts ) || ( nTris && !tris ) || nTris >= ( 1ull << 30 ) )
		return 1;
	g_gpu_build_err[0] = 0;
	if ( int rc = scratch_on_current_device() ) return rc;
	Scratch &S = g_scratch;
	hipStream_t s = (hipStream_t) hipStream;
	if ( *ticket == 0 )
	{
		S.forget_staged();
		S.stageTicket = *ticket = lastTicket.fetch_add( 1 ) + 1;
	}
	else if ( *ticket != S.stageTicket || nVerts < S.stagedV || nTris < S.stagedT )
	{
		// somebody else's geometry lies in the buffers by now, or the arrays shrank: the build will copy everything
		if ( *ticket == S.stageTicket )
			S.forget_staged();
		*ticket = 0;
		return 0;
	}
	if ( int rc = wait_for_staged( S, s ) ) return rc;
	if ( !S.stageDone )
		GCHK( hipEventCreateWithFlags( &S.stageDone, hipEventDisableTiming ) );
	if ( int rc = S.verts.reserve( sizeof( rt_vertex ) * ( nVerts ? nVerts : 1 ), s, sizeof( rt_vertex ) * S.stagedV ) ) return rc;
	if ( int rc = S.tris.reserve( sizeof( rt_triangle ) * ( nTris ? nTris : 1 ), s, sizeof( rt_triangle ) * S.stagedT ) ) return rc;
	if ( nVerts > S.stagedV )
		GCHK( hipMemcpyAsync( S.verts.as<rt_vertex>() + S.stagedV, verts + S.stagedV, sizeof( rt_vertex ) * ( nVerts - S.stagedV ), hipMemcpyHostToDevice, s ) );
	if ( nTris > S.stagedT )
		GCHK( hipMemcpyAsync( S.tris.as<rt_triangle>() + S.stagedT, tris + S.stagedT, sizeof( rt_triangle ) * ( nTris - S.stagedT ), hipMemcpyHostToDevice, s ) );
	S.stagedV = nVerts;
	S.stagedT = nTris;
	GCHK( hipEventRecord( S.stageDone, s ) );
	S.stageStream = s;
	S.stagePending = true;
	return 0;
}

extern "C" void rt_kdtree_device_recycle( void *nodes, void *recs, void *shade, void *leafFull )
{
	void *in[4] = { nodes, recs, shade, leafFull };
	int dev = 0;
	const bool here = hipGetDevice( &dev ) == hipSuccess && ( g_scratchDev == dev || g_scratchDev < 0 );
	if ( here )
		g_scratchDev = dev;
	Scratch &S = g_scratch;
	for ( int k = 0; k < 4; ++k )
	{
		if ( !in[k] )
			continue;
		void *base = nullptr;
		size_t size = 0;
		// (the size as allocated; a pointer into the middle of an allocation, or one this runtime does not know, is freed)
		if ( here && hipMemGetAddressRange( &base, &size, in[k] ) == hipSuccess && base == in[k] && size > S.spareBytes[k] )
		{
			if ( S.spare[k] ) (void) hipFree( S.spare[k] );
			S.spare[k] = in[k];
			S.spareBytes[k] = size;
		}
		else
		{
			(void) hipGetLastError();
			(void) hipFree( in[k] );
		}
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

// the buffers the finished tree is written to (owned by the caller once the build has succeeded)
struct ResultBufs
{
	uint32_t capNodes = 0, capRecs = 0;      // what out->nodes / out->recs (one list) / out->leafFull have room for
	bool two = false;                        // the record array has room for the lists twice, out->leafFull exists
};

static void free_result( rt_kdtree_device *out, ResultBufs &fin )
{
	if ( out->nodes ) (void) hipFree( out->nodes );
	if ( out->recs ) (void) hipFree( out->recs );
	if ( out->leafFull ) (void) hipFree( out->leafFull );
	out->nodes = out->recs = out->leafFull = nullptr;
	fin = ResultBufs();
}

static int alloc_result( rt_kdtree_device *out, size_t nodes, size_t recs, bool two, ResultBufs &fin )
{
	Scratch &S = g_scratch;
	hipError_t ea = S.take( 0, &out->nodes, sizeof( rt_hip_kdnode ) * ( nodes ? nodes : 1 ) );
	if ( ea == hipSuccess ) ea = S.take( 1, &out->recs, sizeof( rt_hip_trirec ) * ( recs * ( two ? 2 : 1 ) + 1 ) );
	if ( ea == hipSuccess && two ) ea = S.take( 3, &out->leafFull, sizeof( rt_hip_leaf_full ) * ( nodes ? nodes : 1 ) );
	if ( ea != hipSuccess )
	{
		snprintf( g_gpu_build_err, sizeof( g_gpu_build_err ), "allocating the tree failed: %s", hipGetErrorString( ea ) );
		(void) hipGetLastError();
		free_result( out, fin );
		return 10;
	}
	fin.capNodes = (uint32_t) nodes;
	fin.capRecs = (uint32_t) recs;
	fin.two = two;
	return 0;
}

// nodes, leaf records (and the pruned lists in front of the complete ones) into the result's buffers - if the tree
// the state describes fits them (k_fit; not at all while there are no buffers)
static int enqueue_result( Scratch &S, const SubTree &sub, BuildState *dState, GeomInfo *dGeom, const ResultBufs &fin, float pruneLeaves, hipStream_t s,
	rt_kdtree_device *out );

static int build_device( const rt_vertex *verts, const rt_triangle *tris, uint64_t vertexCount, uint64_t trianglesCount,
	const rt_kdtree_build_opts *opts, void *hipStream, rt_kdtree_device *out );

extern "C" int rt_kdtree_build_device_ex( const rt_vertex *verts, const rt_triangle *tris, uint64_t vertexCount,
	uint64_t trianglesCount, const rt_kdtree_build_opts *opts, void *hipStream, rt_kdtree_device *out )
{
	g_gpu_build_err[0] = 0;
	if ( !out )
		return 1;
	memset( out, 0, sizeof( *out ) );
	const int rc = build_device( verts, tris, vertexCount, trianglesCount, opts, hipStream, out );
	if ( rc != 0 )
		rt_kdtree_device_release( out );          // (whatever had been allocated for the result by then)
	return rc;
}

static int build_device( const rt_vertex *verts, const rt_triangle *tris, uint64_t vertexCount, uint64_t trianglesCount,
	const rt_kdtree_build_opts *opts, void *hipStream, rt_kdtree_device *out )
{
	if ( !opts )
		return 1;
	if ( opts->sahStop > 0.0f || opts->allAxes )
		return 2;          // (host builder only)
	const int maxDepth = opts->maxDepth, maxPrims = opts->maxPrims;
	if ( !out || ( vertexCount && !verts ) || ( trianglesCount && !tris ) )
		return 1;
	if ( trianglesCount >= ( 1ull << 30 ) || maxDepth < 0 || maxDepth > RT_HIP_MAX_TREE_DEPTH || maxPrims < 1 )
		return 2;
	hipStream_t s = (hipStream_t) hipStream;
	const uint32_t nT = (uint32_t) trianglesCount;

	// Work buffers are kept between builds of one thread on one device (a frame
	// loop rebuilds a tree of about the same size every frame; allocating and
	// freeing them took as long as all the kernels); rt_kdtree_device_trim frees them.
	if ( int rc = scratch_on_current_device() ) return rc;
	const bool TT = getenv( "RT_KD_TIMING" ) != nullptr;      // (development aid: host time at which each phase was reached, to stderr)
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
	// (what rt_kdtree_device_stage has copied already, if it is this build's geometry, stays where it is)
	if ( int rc = wait_for_staged( S, s ) ) return rc;
	uint64_t haveV = 0, haveT = 0;
	if ( opts->stageTicket && opts->stageTicket == S.stageTicket && S.stagedV <= vertexCount && S.stagedT <= nT )
	{
		haveV = S.stagedV;
		haveT = S.stagedT;
	}
	S.forget_staged();
	if ( int rc = S.verts.reserve( sizeof( rt_vertex ) * ( vertexCount ? vertexCount : 1 ), s, sizeof( rt_vertex ) * haveV ) ) return rc;
	if ( int rc = S.tris.reserve( sizeof( rt_triangle ) * ( nT ? nT : 1 ), s, sizeof( rt_triangle ) * haveT ) ) return rc;
	if ( vertexCount > haveV )
		GCHK( hipMemcpyAsync( S.verts.as<rt_vertex>() + haveV, verts + haveV, sizeof( rt_vertex ) * ( vertexCount - haveV ), hipMemcpyHostToDevice, s ) );
	if ( nT > haveT )
		GCHK( hipMemcpyAsync( S.tris.as<rt_triangle>() + haveT, tris + haveT, sizeof( rt_triangle ) * ( nT - haveT ), hipMemcpyHostToDevice, s ) );
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
	const bool wantTwo = opts->pruneLeaves > 0.0f;
	ResultBufs fin;
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
		if ( int rc = S.sub.reserve( ( sizeof( uint4 ) + 3 * sizeof( uint32_t ) ) * (size_t) caps.nodesTotal, s ) ) return rc;
		SubTree sub;
		sub.walk = S.sub.as<uint4>();
		sub.pos = (uint32_t *) ( sub.walk + caps.nodesTotal );
		sub.refPos = sub.pos + caps.nodesTotal;
		sub.longLeaves = sub.refPos + caps.nodesTotal;
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
				LAUNCH( k_hist, dim3( G_REF ), dim3( TB ), 0, s, dState, depth, cur, S.hist.as<uint32_t>(), refs, refNode, Ext{ S.ext.as<float>(), nT } );
			LAUNCH( k_split, dim3( G_NODE * SPLIT_FINER ), dim3( TB ), 0, s, dState, depth, cur, S.hist.as<uint32_t>(), nodeSums );
			LAUNCH( k_children_count, dim3( G_NODE + ( nT ? G_REF : 0 ) ), dim3( TB ), 0, s, dState, depth, cur, next, nodeSums, S.info.as<uint2>(), sub.walk, caps,
				maxDepth, maxPrims, refs, refNode, Ext{ S.ext.as<float>(), nT }, refSums );
			if ( nT )
				LAUNCH( k_scatter, dim3( G_REF ), dim3( TB ), 0, s, dState, depth, cur, next, refs, refNode, Ext{ S.ext.as<float>(), nT }, refSums,
					nRefs, nRefNode, S.leafPool.as<uint32_t>(), S.hist.as<uint32_t>(), &dGeom->lostInPlane );
		}
		// pre-order positions
		LAUNCH( k_walk, dim3( G_FIN ), dim3( TB ), 0, s, dState, sub );
		stamp( "levels enqueued" );

		// While all that runs: the result's buffers, sized by what the previous build of as many triangles came to
		// (a frame loop rebuilds about the same tree every frame; allocating 0.4 GB takes 0.3 ms, a twelfth of a build).
		if ( !out->shade )
		{
			hipError_t ea = S.take( 2, &out->shade, sizeof( rt_hip_trishade ) * ( nT ? nT : 1 ) );
			if ( ea != hipSuccess )
			{
				snprintf( g_gpu_build_err, sizeof( g_gpu_build_err ), "allocating the tree failed: %s", hipGetErrorString( ea ) );
				return 10;
			}
			if ( nT )
				LAUNCH( k_shade_records, dim3( grid_of( nT ) ), dim3( TB ), 0, s, dGeom, S.verts.as<rt_vertex>(), S.tris.as<rt_triangle>(), nT,
					(rt_hip_trishade *) out->shade );
			if ( S.lastNT == nT )
				if ( int rc = alloc_result( out, (size_t) S.lastNodes + S.lastNodes / 8 + 1024, (size_t) S.lastRecs + S.lastRecs / 8 + 1024, wantTwo, fin ) )
				{
					(void) rc;                       // (the exact sizes will be tried once they are known)
					g_gpu_build_err[0] = 0;
				}
		}
		// ... and the tree into them, if it turns out to fit
		if ( int rc = enqueue_result( S, sub, dState, dGeom, fin, opts->pruneLeaves, s, out ) ) return rc;
		GCHK( hipMemcpyAsync( hState, dState, sizeof( BuildState ), hipMemcpyDeviceToHost, s ) );
		GCHK( hipMemcpyAsync( hGeom, dGeom, sizeof( GeomInfo ), hipMemcpyDeviceToHost, s ) );
		GCHK( hipStreamSynchronize( s ) );
		stamp( "levels done" );
		S.histClean = true;
		if ( hGeom->badIndex )
			return 3;
		if ( hState->overflow == OVF_NONE )
		{
			if ( hState->nLevels < 1 || (int) hState->nLevels > maxDepth + 1 || hState->nodesTotal == 0 )
			{
				snprintf( g_gpu_build_err, sizeof( g_gpu_build_err ), "kd build left an inconsistent state (%u levels, %u nodes)", hState->nLevels,
					hState->nodesTotal );
				return 10;
			}
			if ( wantTwo && 2ull * hState->leafRefs >= ( 1ull << 30 ) )
			{
				snprintf( g_gpu_build_err, sizeof( g_gpu_build_err ), "pruned and complete leaf lists together exceed the 30-bit record index" );
				return 4;
			}
			if ( !hState->fits )
			{
				// the first build of this size, or one that outgrew the guess: now that the sizes are known
				free_result( out, fin );
				if ( int rc = alloc_result( out, hState->nodesTotal, hState->leafRefs, wantTwo, fin ) ) return rc;
				stamp( "outputs allocated" );
				if ( int rc = enqueue_result( S, sub, dState, dGeom, fin, opts->pruneLeaves, s, out ) ) return rc;
				GCHK( hipMemcpyAsync( hState, dState, sizeof( BuildState ), hipMemcpyDeviceToHost, s ) );
				GCHK( hipStreamSynchronize( s ) );
				if ( !hState->fits )
				{
					snprintf( g_gpu_build_err, sizeof( g_gpu_build_err ), "kd build: the tree does not fit the buffers allocated for it" );
					return 10;
				}
			}
			break;
		}
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
	const uint32_t nNodes = hState->nodesTotal, nRecs = hState->leafRefs, rootDepth = hState->rootDepth;
	const bool twoLists = wantTwo && nRecs != 0u;
	if ( out->leafFull && !twoLists )
	{
		(void) hipFree( out->leafFull );
		out->leafFull = nullptr;
	}
	stamp( "done" );
	S.lastNT = nT;
	S.lastNodes = nNodes;
	S.lastRecs = nRecs;
	out->nNodes = nNodes;
	out->nRecs = nRecs;
	out->nRecsFull = twoLists ? nRecs : 0;
	out->nTris = nT;
	out->depth = (int) rootDepth;
	return 0;
}

static int enqueue_result( Scratch &S, const SubTree &sub, BuildState *dState, GeomInfo *dGeom, const ResultBufs &fin, float pruneLeaves, hipStream_t s,
	rt_kdtree_device *out )
{
	LAUNCH( k_fit, dim3( 1 ), dim3( 1 ), 0, s, dState, dGeom, fin.capNodes, fin.capRecs, fin.two ? 1u : 0u );
	if ( !out->nodes )
		return 0;
	const uint2 *info = S.info.as<uint2>();
	rt_hip_kdnode *nodes = (rt_hip_kdnode *) out->nodes;
	rt_hip_trirec *recs = (rt_hip_trirec *) out->recs;
	if ( int rc = S.order.reserve( sizeof( uint32_t ) * ( (size_t) fin.capRecs + 1 ), s ) ) return rc;
	LAUNCH( k_emit, dim3( G_FIN ), dim3( TB ), 0, s, dState, info, sub, S.leafPool.as<uint32_t>(), nodes, S.order.as<uint32_t>() );
	LAUNCH( k_emit_long, dim3( G_FIN ), dim3( TB ), 0, s, dState, info, sub, S.leafPool.as<uint32_t>(), S.order.as<uint32_t>() );
	LAUNCH( k_records, dim3( 4 * G_FIN ), dim3( TB ), 0, s, dState, S.verts.as<rt_vertex>(), S.tris.as<rt_triangle>(), S.order.as<uint32_t>(), recs );
	if ( fin.two )
	{
		// pruned lists: the complete ones are kept behind them (read by rays parallel to an axis, see rt_kdtree.h)
		LAUNCH( k_leaf_full, dim3( grid_of( fin.capNodes ) ), dim3( TB ), 0, s, dState, nodes, (rt_hip_leaf_full *) out->leafFull );
		LAUNCH( k_copy_full, dim3( 4 * G_FIN ), dim3( TB ), 0, s, dState, recs );
		LAUNCH( k_prune_leaves, dim3( grid_of( fin.capNodes ) ), dim3( TB ), 0, s, dState, nodes, recs, dGeom, pruneLeaves );
	}
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
