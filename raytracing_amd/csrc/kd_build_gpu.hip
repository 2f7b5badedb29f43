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
#include <hipcub/hipcub.hpp>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rt_kdtree.h"

namespace {

constexpr int PARTS = SAH_PARTS;
constexpr int TB = 256;

#define GCHK( x ) do { hipError_t e_ = ( x ); if ( e_ != hipSuccess ) { \
	snprintf( g_gpu_build_err, sizeof( g_gpu_build_err ), "%s failed: %s (%s:%d)", #x, hipGetErrorString( e_ ), __FILE__, __LINE__ ); \
	return 10; } } while ( 0 )

thread_local char g_gpu_build_err[512] = "";

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

// one wave: folds the partial boxes, applies the rules for vertex 0 and the
// margin, and fills the root's entry of the level table
__global__ __launch_bounds__( 64 ) void k_box_final( const rt_vertex *verts, uint64_t nV, const float *part, int nPart, uint32_t nT,
	GeomInfo *info, float *rootBox, uint32_t *rootStart, uint32_t *rootCount )
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
		rootBox[a] = c;
		rootBox[3 + a] = e;
	}
	*rootStart = 0u;
	*rootCount = nT;
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

// ---- level tables (one entry per node of the level) ----------------------
struct LevelTab
{
	float *box;            // [6 * n]: centre xyz, half extent xyz
	uint32_t *start, *count;
};

struct NodeWork            // per node of the current level
{
	uint32_t *split;       // 1 = inner node
	uint32_t *axis;
	uint32_t *sorted;      // candidate planes ascending (always, except for NaN / inverted boxes)
	float *mn, *delta;
	uint32_t *hist;        // [66 * n]: sorted: histL[33], histR[33]; else L[32], R[32] directly
	float *sep;
	uint32_t *cntL, *cntR;
	uint32_t *childRefs, *leafCnt;
	struct Trip *incl;     // inclusive scan of (split, childRefs, leafCnt), one pass for the three
	__device__ uint32_t splitIdx( uint32_t i ) const;
	__device__ uint32_t childStart( uint32_t i ) const;
	__device__ uint32_t leafStart( uint32_t i ) const;
};

struct alignas( 16 ) Trip { uint32_t a, b, c, d; };
struct TripSum
{
	__host__ __device__ Trip operator()( const Trip &x, const Trip &y ) const { return Trip{ x.a + y.a, x.b + y.b, x.c + y.c, 0u }; }
};
struct TripLoad
{
	const uint32_t *a, *b, *c;
	__host__ __device__ Trip operator()( uint32_t i ) const { return Trip{ a[i], b[i], c[i], 0u }; }
};
// exclusive values = inclusive value - own contribution
__device__ inline uint32_t NodeWork::splitIdx( uint32_t i ) const { return incl[i].a - split[i]; }
__device__ inline uint32_t NodeWork::childStart( uint32_t i ) const { return incl[i].b - childRefs[i]; }
__device__ inline uint32_t NodeWork::leafStart( uint32_t i ) const { return incl[i].c - leafCnt[i]; }

__device__ inline float box_c( const float *b, int ax ) { return b[ax]; }
__device__ inline float box_e( const float *b, int ax ) { return b[3 + ax]; }

__global__ __launch_bounds__( TB ) void k_decide( LevelTab L, NodeWork W, uint32_t n, int depth, int maxDepth, int maxPrims )
{
	const uint32_t i = blockIdx.x * TB + threadIdx.x;
	if ( i >= n )
		return;
	const float *b = L.box + 6ull * i;
	const uint32_t cnt = L.count[i];
	for ( int k = 0; k < 66; ++k )
		W.hist[66ull * i + k] = 0u;
	if ( cnt <= (uint32_t) maxPrims || depth == maxDepth )
	{
		W.split[i] = 0u;
		W.leafCnt[i] = cnt;
		W.childRefs[i] = 0u;
		return;
	}
	// longest half extent, ties X > Y > Z (:945-957)
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
	W.split[i] = 1u;
	W.leafCnt[i] = 0u;
	W.axis[i] = (uint32_t) ax;
	W.sorted[i] = sorted ? 1u : 0u;
	W.mn[i] = mn;
	W.delta[i] = delta;
}

// Counting goes through a histogram in LDS for the node the block's first
// reference belongs to (near the root a node spans thousands of blocks, and
// atomics of all of them on the same 66 words of HBM would be executed one
// after the other); references of other nodes inside the block count directly.
__global__ __launch_bounds__( TB ) void k_hist( NodeWork W, const uint32_t *refs, const uint32_t *refNode, uint32_t nRefs,
	const float *ext, uint32_t nT )
{
	__shared__ uint32_t sh[66];
	const uint32_t j = blockIdx.x * TB + threadIdx.x;
	const uint32_t node0 = refNode[blockIdx.x * TB];
	if ( threadIdx.x < 66 )
		sh[threadIdx.x] = 0u;
	__syncthreads();
	if ( j < nRefs )
	{
		const uint32_t i = refNode[j];
		if ( W.split[i] )
		{
			const uint32_t ax = W.axis[i], tri = refs[j];
			const float a = ext[(size_t) ax * nT + tri], b = ext[( 3ull + ax ) * nT + tri];
			const float mn = W.mn[i], delta = W.delta[i];
			uint32_t *h = i == node0 ? sh : W.hist + 66ull * i;
			if ( W.sorted[i] )
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
				atomicAdd( &h[lo], 1u );
				atomicAdd( &h[33 + hi], 1u );
			}
			else
			{
				for ( int k = 0; k < PARTS; ++k )
				{
					const float s = mn + k * delta;
					if ( a < s ) atomicAdd( &h[k], 1u );
					if ( b > s ) atomicAdd( &h[33 + k], 1u );
				}
			}
		}
	}
	__syncthreads();
	if ( threadIdx.x < 66 && sh[threadIdx.x] )
		atomicAdd( &W.hist[66ull * node0 + threadIdx.x], sh[threadIdx.x] );
}

// SAH sweep of :591-857 on the counts: lowest index with the strictly smallest cost
__global__ __launch_bounds__( TB ) void k_split( LevelTab L, NodeWork W, uint32_t n )
{
	const uint32_t i = blockIdx.x * TB + threadIdx.x;
	if ( i >= n || !W.split[i] )
		return;
	const float *b = L.box + 6ull * i;
	const int ax = (int) W.axis[i];
	const float c = box_c( b, ax ), e = box_e( b, ax );
	const float mn = W.mn[i], delta = W.delta[i];
	float eA, eB;
	if ( ax == 0 ) { eA = b[4]; eB = b[5]; }
	else if ( ax == 1 ) { eA = b[3]; eB = b[5]; }
	else { eA = b[3]; eB = b[4]; }
	const float area = eA * eB;
	const uint32_t *h = W.hist + 66ull * i;
	const bool sorted = W.sorted[i] != 0u;

	// R[k] needs a suffix sum: total first
	unsigned long long accR = 0;
	if ( sorted )
		for ( int k = 1; k <= PARTS; ++k ) accR += h[33 + k];
	unsigned long long accL = 0;
	float best = 0.0f, outSep = 0.0f;
	unsigned long long cl = 0, cr = 0;
	for ( int k = 0; k < PARTS; ++k )
	{
		unsigned long long Lk, Rk;
		if ( sorted )
		{
			accL += h[k];
			Lk = accL;
			Rk = accR;               // sum of histR[k+1 .. 32]
			accR -= h[33 + k + 1];
		}
		else
		{
			Lk = h[k];
			Rk = h[33 + k];
		}
		const float sep = mn + k * delta;
		const float sah = ( area + ( sep - c + e ) * ( eA + eB ) ) * Lk + ( area + ( c + e - sep ) * ( eA + eB ) ) * Rk;
		if ( k == 0 || sah < best )
		{
			best = sah;
			outSep = sep;
			cl = Lk;
			cr = Rk;
		}
	}
	W.sep[i] = outSep;
	W.cntL[i] = (uint32_t) cl;
	W.cntR[i] = (uint32_t) cr;
	W.childRefs[i] = (uint32_t) ( cl + cr );
}

// child boxes: slab split at sep (:1034-1073); record of the node in level order
__global__ __launch_bounds__( TB ) void k_children( LevelTab L, NodeWork W, LevelTab N, uint32_t n, uint2 *info, uint32_t idBase,
	uint32_t nextBase, uint32_t leafPoolBase )
{
	const uint32_t i = blockIdx.x * TB + threadIdx.x;
	if ( i >= n )
		return;
	if ( !W.split[i] )
	{
		info[idBase + i] = make_uint2( leafPoolBase + W.leafStart( i ), ( L.count[i] << 2 ) | 3u );
		return;
	}
	const float *b = L.box + 6ull * i;
	const int ax = (int) W.axis[i];
	const float sep = W.sep[i];
	const uint32_t c = 2u * W.splitIdx( i );
	float *bl = N.box + 6ull * c, *br = bl + 6;
	for ( int k = 0; k < 6; ++k )
		bl[k] = br[k] = b[k];
	bl[ax] = 0.5f * ( box_c( b, ax ) - box_e( b, ax ) + sep );
	bl[3 + ax] = 0.5f * ( sep - box_c( b, ax ) + box_e( b, ax ) );
	br[ax] = 0.5f * ( box_c( b, ax ) + box_e( b, ax ) + sep );
	br[3 + ax] = 0.5f * ( box_c( b, ax ) + box_e( b, ax ) - sep );
	N.start[c] = W.childStart( i );
	N.count[c] = W.cntL[i];
	N.start[c + 1] = W.childStart( i ) + W.cntL[i];
	N.count[c + 1] = W.cntR[i];
	info[idBase + i] = make_uint2( __float_as_uint( sep ), ( ( nextBase + c ) << 2 ) | (uint32_t) ax );
}

// per reference: goes left (low word) / right (high word); evaluated on the
// fly by the scan and again by the scatter, never stored
struct RefFlags
{
	NodeWork W;
	const uint32_t *refs, *refNode;
	const float *ext;
	uint32_t nT;
	__host__ __device__ unsigned long long operator()( uint32_t j ) const
	{
		const uint32_t i = refNode[j];
		unsigned long long f = 0ull;
		if ( W.split[i] )
		{
			const uint32_t ax = W.axis[i], tri = refs[j];
			const float sep = W.sep[i];
			if ( ext[(size_t) ax * nT + tri] < sep ) f |= 1ull;
			if ( ext[( 3ull + ax ) * nT + tri] > sep ) f |= 1ull << 32;
		}
		return f;
	}
};

// stable partition (:971-1029): the k-th reference of a node that goes left
// becomes the k-th reference of its left child; leaves move to the leaf pool
__global__ __launch_bounds__( TB ) void k_scatter( LevelTab L, RefFlags F, uint32_t nRefs,
	const unsigned long long *scan, uint32_t *nextRefs, uint32_t *nextRefNode,
	uint32_t *leafPool, uint32_t leafPoolBase, uint32_t *lostInPlane )
{
	const uint32_t j = blockIdx.x * TB + threadIdx.x;
	if ( j >= nRefs )
		return;
	const NodeWork &W = F.W;
	const uint32_t i = F.refNode[j], tri = F.refs[j], s0 = L.start[i];
	if ( !W.split[i] )
	{
		leafPool[leafPoolBase + W.leafStart( i ) + ( j - s0 )] = tri;
		return;
	}
	const unsigned long long f = F( j ), here = scan[j], base = scan[s0];
	const uint32_t c = 2u * W.splitIdx( i );
	if ( f == 0ull )
		atomicAdd( lostInPlane, 1u );             // (rare: a triangle flat in the split axis that lies in the plane)
	if ( f & 1ull )
	{
		const uint32_t p = W.childStart( i ) + ( (uint32_t) here - (uint32_t) base );
		nextRefs[p] = tri;
		nextRefNode[p] = c;
	}
	if ( f >> 32 )
	{
		const uint32_t p = W.childStart( i ) + W.cntL[i] + ( (uint32_t) ( here >> 32 ) - (uint32_t) ( base >> 32 ) );
		nextRefs[p] = tri;
		nextRefNode[p] = c + 1u;
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
	DVec verts, tris, ext, refsA, refsB, nodeA, nodeB, scan, cub, info, leafPool, work, sub, order, geom;
	uint32_t *hTotals = nullptr;
	void release()
	{
		for ( DVec *v : { &verts, &tris, &ext, &refsA, &refsB, &nodeA, &nodeB, &scan, &cub, &info, &leafPool, &work, &sub, &order, &geom } )
			v->release();
		if ( hTotals ) (void) hipHostFree( hTotals );
		hTotals = nullptr;
	}
};
thread_local Scratch g_scratch;
thread_local int g_scratchDev = -1;

// carve the per-level node tables out of one allocation
int level_tab( DVec &buf, size_t n, hipStream_t s, LevelTab &t )
{
	if ( n == 0 ) n = 1;
	if ( int rc = buf.reserve( n * ( 6 * sizeof( float ) + 2 * sizeof( uint32_t ) ), s ) ) return rc;
	t.box = buf.as<float>();
	t.start = (uint32_t *) ( t.box + 6 * n );
	t.count = t.start + n;
	return 0;
}

int node_work( DVec &buf, size_t n, hipStream_t s, NodeWork &w )
{
	if ( n == 0 ) n = 1;
	if ( int rc = buf.reserve( n * ( 4 + 66 + 11 ) * sizeof( uint32_t ), s ) ) return rc;
	uint32_t *p = buf.as<uint32_t>();
	w.incl = (Trip *) p; p += 4 * n;          // first: 16-byte aligned
	w.hist = p; p += 66 * n;
	w.split = p; p += n;
	w.axis = p; p += n;
	w.sorted = p; p += n;
	w.mn = (float *) p; p += n;
	w.delta = (float *) p; p += n;
	w.sep = (float *) p; p += n;
	w.cntL = p; p += n;
	w.cntR = p; p += n;
	w.childRefs = p; p += n;
	w.leafCnt = p; p += n;
	return 0;
}

template <class In, class T>
int exclusive_sum( DVec &cub, In in, T *out, size_t n, hipStream_t s )
{
	size_t bytes = 0;
	GCHK( hipcub::DeviceScan::ExclusiveSum( nullptr, bytes, in, out, (int) n, s ) );
	if ( int rc = cub.reserve( bytes ? bytes : 16, s ) ) return rc;
	GCHK( hipcub::DeviceScan::ExclusiveSum( cub.p, bytes, in, out, (int) n, s ) );
	return 0;
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
	Scratch &S = g_scratch;
	if ( !S.hTotals )
		GCHK( hipHostMalloc( (void **) &S.hTotals, 4 * sizeof( uint32_t ) + sizeof( GeomInfo ), hipHostMallocDefault ) );
	uint32_t *hTotals = S.hTotals;
	GeomInfo *hGeom = (GeomInfo *) ( S.hTotals + 4 );

	// inputs; the index check, the highest material slot and the scene box are
	// taken from the device copy (a host pass over 1 M triangles and their
	// vertices costs 2 ms, more than a quarter of the whole build)
	if ( int rc = S.verts.reserve( sizeof( rt_vertex ) * ( vertexCount ? vertexCount : 1 ), s ) ) return rc;
	if ( int rc = S.tris.reserve( sizeof( rt_triangle ) * ( nT ? nT : 1 ), s ) ) return rc;
	if ( vertexCount ) GCHK( hipMemcpyAsync( S.verts.p, verts, sizeof( rt_vertex ) * vertexCount, hipMemcpyHostToDevice, s ) );
	if ( nT ) GCHK( hipMemcpyAsync( S.tris.p, tris, sizeof( rt_triangle ) * nT, hipMemcpyHostToDevice, s ) );
	if ( int rc = S.ext.reserve( 6 * sizeof( float ) * ( nT ? nT : 1 ), s ) ) return rc;
	if ( int rc = S.geom.reserve( sizeof( GeomInfo ) + 6 * sizeof( float ) * BOX_BLOCKS, s ) ) return rc;
	GeomInfo *dGeom = S.geom.as<GeomInfo>();
	float *boxPart = (float *) ( dGeom + 1 );
	GCHK( hipMemsetAsync( dGeom, 0, sizeof( GeomInfo ), s ) );
	if ( nT ) hipLaunchKernelGGL( k_extents, dim3( grid_of( nT ) ), dim3( TB ), 0, s, S.verts.as<rt_vertex>(), S.tris.as<rt_triangle>(), nT,
		vertexCount, S.ext.as<float>(), dGeom );

	// level 0: the root holds every triangle in index order
	DVec *refsCur = &S.refsA, *refsNext = &S.refsB, *tabCur = &S.nodeA, *tabNext = &S.nodeB;
	if ( int rc = refsCur->reserve( 2 * sizeof( uint32_t ) * ( nT ? nT : 1 ), s ) ) return rc;
	size_t refCap = refsCur->cap / ( 2 * sizeof( uint32_t ) );
	uint32_t *refs = refsCur->as<uint32_t>(), *refNode = refs + refCap;
	if ( nT ) hipLaunchKernelGGL( k_iota, dim3( grid_of( nT ) ), dim3( TB ), 0, s, refs, refNode, nT );
	LevelTab cur;
	if ( int rc = level_tab( *tabCur, 1, s, cur ) ) return rc;
	{
		const uint64_t want = ( vertexCount + TB - 1 ) / TB;
		const int nPart = (int) ( want < 1 ? 1 : want > (uint64_t) BOX_BLOCKS ? (uint64_t) BOX_BLOCKS : want );
		hipLaunchKernelGGL( k_box_partial, dim3( nPart ), dim3( TB ), 0, s, S.verts.as<rt_vertex>(), vertexCount, boxPart );
		hipLaunchKernelGGL( k_box_final, dim3( 1 ), dim3( 64 ), 0, s, S.verts.as<rt_vertex>(), vertexCount, boxPart, nPart, nT, dGeom,
			cur.box, cur.start, cur.count );
		GCHK( hipMemcpyAsync( hGeom, dGeom, sizeof( GeomInfo ), hipMemcpyDeviceToHost, s ) );
		GCHK( hipStreamSynchronize( s ) );
		if ( hGeom->badIndex )
			return 3;
		out->maxMaterial = hGeom->maxMat;
		out->bbox.center.x = hGeom->box[0]; out->bbox.center.y = hGeom->box[1]; out->bbox.center.z = hGeom->box[2];
		out->bbox.extents.x = hGeom->box[3]; out->bbox.extents.y = hGeom->box[4]; out->bbox.extents.z = hGeom->box[5];
	}

	std::vector<uint32_t> levelFirst, levelCount;
	uint64_t nodesTotal = 0, leafRefs = 0;
	uint32_t levelNodes = 1, levelRefs = nT;
	for ( int depth = 0; ; ++depth )
	{
		levelFirst.push_back( (uint32_t) nodesTotal );
		levelCount.push_back( levelNodes );
		if ( nodesTotal + levelNodes >= ( 1ull << 30 ) )
			return 4;
		NodeWork W;
		if ( int rc = node_work( S.work, levelNodes, s, W ) ) return rc;
		if ( int rc = S.info.reserve( sizeof( uint2 ) * ( nodesTotal + levelNodes ), s, sizeof( uint2 ) * nodesTotal ) ) return rc;

		hipLaunchKernelGGL( k_decide, dim3( grid_of( levelNodes ) ), dim3( TB ), 0, s, cur, W, levelNodes, depth, maxDepth, maxPrims );
		if ( levelRefs )
			hipLaunchKernelGGL( k_hist, dim3( grid_of( levelRefs ) ), dim3( TB ), 0, s, W, refs, refNode, levelRefs, S.ext.as<float>(), nT );
		hipLaunchKernelGGL( k_split, dim3( grid_of( levelNodes ) ), dim3( TB ), 0, s, cur, W, levelNodes );
		{
			hipcub::TransformInputIterator<Trip, TripLoad, hipcub::CountingInputIterator<uint32_t>> in(
				hipcub::CountingInputIterator<uint32_t>( 0u ), TripLoad{ W.split, W.childRefs, W.leafCnt } );
			size_t bytes = 0;
			GCHK( hipcub::DeviceScan::InclusiveScan( nullptr, bytes, in, W.incl, TripSum(), (int) levelNodes, s ) );
			if ( int rc = S.cub.reserve( bytes ? bytes : 16, s ) ) return rc;
			GCHK( hipcub::DeviceScan::InclusiveScan( S.cub.p, bytes, in, W.incl, TripSum(), (int) levelNodes, s ) );
		}
		// the last inclusive value holds the three totals
		GCHK( hipMemcpyAsync( hTotals, W.incl + ( levelNodes - 1 ), 3 * sizeof( uint32_t ), hipMemcpyDeviceToHost, s ) );
		GCHK( hipStreamSynchronize( s ) );
		const uint32_t splits = hTotals[0], childRefs = hTotals[1], levelLeafRefs = hTotals[2];
		if ( (uint64_t) splits * 2 + nodesTotal + levelNodes >= ( 1ull << 30 ) || leafRefs + levelLeafRefs + childRefs >= ( 1ull << 30 ) )
			return 4;          // 30-bit links of the 8-byte node

		// next level's tables and reference arrays, leaf pool
		LevelTab next;
		if ( int rc = level_tab( *tabNext, 2 * (size_t) splits, s, next ) ) return rc;
		if ( int rc = refsNext->reserve( 2 * sizeof( uint32_t ) * ( childRefs ? childRefs : 1 ), s ) ) return rc;
		const size_t nextCap = refsNext->cap / ( 2 * sizeof( uint32_t ) );
		uint32_t *nRefs = refsNext->as<uint32_t>(), *nRefNode = nRefs + nextCap;
		if ( int rc = S.leafPool.reserve( sizeof( uint32_t ) * ( leafRefs + levelLeafRefs + 1 ), s, sizeof( uint32_t ) * leafRefs ) ) return rc;

		hipLaunchKernelGGL( k_children, dim3( grid_of( levelNodes ) ), dim3( TB ), 0, s, cur, W, next, levelNodes, S.info.as<uint2>(),
			(uint32_t) nodesTotal, (uint32_t) ( nodesTotal + levelNodes ), (uint32_t) leafRefs );
		if ( levelRefs )
		{
			if ( int rc = S.scan.reserve( sizeof( unsigned long long ) * levelRefs, s ) ) return rc;
			const RefFlags F{ W, refs, refNode, S.ext.as<float>(), nT };
			hipcub::TransformInputIterator<unsigned long long, RefFlags, hipcub::CountingInputIterator<uint32_t>> in(
				hipcub::CountingInputIterator<uint32_t>( 0u ), F );
			if ( int rc = exclusive_sum( S.cub, in, S.scan.as<unsigned long long>(), levelRefs, s ) ) return rc;
			hipLaunchKernelGGL( k_scatter, dim3( grid_of( levelRefs ) ), dim3( TB ), 0, s, cur, F, levelRefs,
				S.scan.as<unsigned long long>(), nRefs, nRefNode, S.leafPool.as<uint32_t>(), (uint32_t) leafRefs, &dGeom->lostInPlane );
		}

		nodesTotal += levelNodes;
		leafRefs += levelLeafRefs;
		if ( splits == 0 )
			break;
		std::swap( refsCur, refsNext );
		std::swap( tabCur, tabNext );
		cur = next;
		refs = nRefs;
		refNode = nRefNode;
		levelNodes = 2 * splits;
		levelRefs = childRefs;
	}

	// sizes bottom-up, positions top-down
	const uint32_t nNodes = (uint32_t) nodesTotal;
	if ( int rc = S.sub.reserve( 5 * sizeof( uint32_t ) * (size_t) nNodes, s ) ) return rc;
	uint32_t *subNodes = S.sub.as<uint32_t>(), *subRefs = subNodes + nNodes, *subDepth = subRefs + nNodes, *pos = subDepth + nNodes, *refPos = pos + nNodes;
	const uint2 *info = S.info.as<uint2>();
	for ( int d = (int) levelFirst.size() - 1; d >= 0; --d )
		hipLaunchKernelGGL( k_sizes, dim3( grid_of( levelCount[d] ) ), dim3( TB ), 0, s, info, levelFirst[d], levelCount[d], subNodes, subRefs, subDepth );
	GCHK( hipMemsetAsync( pos, 0, sizeof( uint32_t ), s ) );
	GCHK( hipMemsetAsync( refPos, 0, sizeof( uint32_t ), s ) );
	for ( size_t d = 0; d < levelFirst.size(); ++d )
		hipLaunchKernelGGL( k_positions, dim3( grid_of( levelCount[d] ) ), dim3( TB ), 0, s, info, levelFirst[d], levelCount[d], subNodes, subRefs, pos, refPos );
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
	if ( int rc = S.order.reserve( sizeof( uint32_t ) * ( (size_t) nRecs + 1 ), s ) ) { rt_kdtree_device_release( out ); return rc; }
	hipLaunchKernelGGL( k_emit, dim3( grid_of( nNodes ) ), dim3( TB ), 0, s, info, nNodes, pos, refPos, S.leafPool.as<uint32_t>(),
		(rt_hip_kdnode *) out->nodes, S.order.as<uint32_t>() );
	hipLaunchKernelGGL( k_records, dim3( grid_of( (size_t) nRecs + 1 ) ), dim3( TB ), 0, s, S.verts.as<rt_vertex>(), S.tris.as<rt_triangle>(),
		S.order.as<uint32_t>(), nRecs, (rt_hip_trirec *) out->recs );
	if ( nT )
		hipLaunchKernelGGL( k_shade_records, dim3( grid_of( nT ) ), dim3( TB ), 0, s, S.verts.as<rt_vertex>(), S.tris.as<rt_triangle>(), nT,
			(rt_hip_trishade *) out->shade );
	if ( twoLists )
	{
		rt_hip_trirec *recs = (rt_hip_trirec *) out->recs;
		hipLaunchKernelGGL( k_leaf_full, dim3( grid_of( nNodes ) ), dim3( TB ), 0, s, (const rt_hip_kdnode *) out->nodes, nNodes, nRecs,
			(rt_hip_leaf_full *) out->leafFull );
		GCHK( hipMemcpyAsync( recs + nRecs, recs, sizeof( rt_hip_trirec ) * (size_t) nRecs, hipMemcpyDeviceToDevice, s ) );
		GCHK( hipMemsetAsync( recs + 2 * (size_t) nRecs, 0, sizeof( rt_hip_trirec ), s ) );       // the spare record at the very end
		hipLaunchKernelGGL( k_prune_leaves, dim3( grid_of( nNodes ) ), dim3( TB ), 0, s, (rt_hip_kdnode *) out->nodes, nNodes,
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
