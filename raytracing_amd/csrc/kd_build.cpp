// Host kd-tree builder (see include/rt_kdtree.h for what it keeps from the
// reference's builder, rt_funcs_render_pipe.c:591-1209, and what it changes).
//
// All fp32 expressions are written in the reference's operand order and this
// file is compiled with -ffp-contract=off: the split positions, counts and
// child boxes, and therefore the tree, are the reference's bit for bit.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include <unistd.h>

#include "rt_funcs_render_pipe.h"
#include "rt_kdtree.h"

namespace {

inline float box_c( const rt_box &b, int ax ) { return ax == 0 ? b.center.x : ax == 1 ? b.center.y : b.center.z; }
inline float box_e( const rt_box &b, int ax ) { return ax == 0 ? b.extents.x : ax == 1 ? b.extents.y : b.extents.z; }
inline void box_set( rt_box &b, int ax, float c, float e )
{
	if ( ax == 0 ) { b.center.x = c; b.extents.x = e; }
	else if ( ax == 1 ) { b.center.y = c; b.extents.y = e; }
	else { b.center.z = c; b.extents.z = e; }
}

// extent of one triangle on one axis: the compare chain of the reference
// (rt_funcs_render_pipe.c:620-633)
inline void tri_extent( float a, float b, float c, float &mn, float &mx )
{
	mx = a;
	if ( mx < b ) mx = b;
	if ( mx < c ) mx = c;
	mn = a;
	if ( mn > b ) mn = b;
	if ( mn > c ) mn = c;
}

// longest half extent, ties X > Y > Z (:945-957)
inline int split_axis( const rt_box &b )
{
	float range = b.extents.x;
	int ax = RT_AXIS_X;
	if ( range < b.extents.y ) { range = b.extents.y; ax = RT_AXIS_Y; }
	if ( range < b.extents.z ) { range = b.extents.z; ax = RT_AXIS_Z; }
	return ax;
}

// child boxes: slab split at sep (:1034-1073)
inline void child_boxes( const rt_box &b, int ax, float sep, rt_box &l, rt_box &r )
{
	l = r = b;
	box_set( l, ax, 0.5f * ( box_c( b, ax ) - box_e( b, ax ) + sep ), 0.5f * ( sep - box_c( b, ax ) + box_e( b, ax ) ) );
	box_set( r, ax, 0.5f * ( box_c( b, ax ) + box_e( b, ax ) + sep ), 0.5f * ( box_c( b, ax ) + box_e( b, ax ) - sep ) );
}

// The SAH sweep of :591-857 for triangles given by their extents on the split
// axis.  get(j, mn, mx) yields the extent of the node's j-th triangle.
// Candidate i is sep_i = min + i * delta; NL_i = #(mn < sep_i), NR_i = #(mx > sep_i);
// the winner is the lowest i with the strictly smallest cost.
template <class Get>
void sah_sweep( uint64_t n, int ax, const rt_box &box, Get get, uint64_t &cntL, uint64_t &cntR, float &outSep,
	float *bestCost = nullptr, float *leafCost = nullptr )
{
	const float c = box_c( box, ax ), e = box_e( box, ax );
	const float mn = c - e, mx = c + e;
	const float delta = ( mx - mn ) / (float) SAH_PARTS;
	float eA, eB;
	if ( ax == 0 ) { eA = box.extents.y; eB = box.extents.z; }
	else if ( ax == 1 ) { eA = box.extents.x; eB = box.extents.z; }
	else { eA = box.extents.x; eB = box.extents.y; }
	const float area = eA * eB;

	float seps[SAH_PARTS];
	bool sorted = true;
	for ( int i = 0; i < SAH_PARTS; ++i )
	{
		seps[i] = mn + i * delta;
		if ( i && !( seps[i - 1] <= seps[i] ) ) sorted = false;
	}

	uint64_t L[SAH_PARTS], R[SAH_PARTS];
	if ( sorted )
	{
		// one pass: histL[k] = #triangles whose first candidate above mn is k,
		// histR[k] = #triangles with exactly k candidates below mx
		uint64_t histL[SAH_PARTS + 1] = { 0 }, histR[SAH_PARTS + 1] = { 0 };
		for ( uint64_t j = 0; j < n; ++j )
		{
			float a, b;
			get( j, a, b );
			int lo = 0, hi = 0;
			for ( int step = SAH_PARTS / 2; step; step >>= 1 )      // seps has 32 entries: 5 steps + 1
			{
				if ( !( a < seps[lo + step - 1] ) ) lo += step;
				if ( b > seps[hi + step - 1] ) hi += step;
			}
			if ( lo < SAH_PARTS && !( a < seps[lo] ) ) ++lo;
			if ( hi < SAH_PARTS && b > seps[hi] ) ++hi;
			++histL[lo];
			++histR[hi];
		}
		uint64_t acc = 0;
		for ( int i = 0; i < SAH_PARTS; ++i ) { acc += histL[i]; L[i] = acc; }
		acc = 0;
		for ( int i = SAH_PARTS - 1; i >= 0; --i ) { acc += histR[i + 1]; R[i] = acc; }
	}
	else
	{
		// degenerate box (NaN / inverted): literal 32-pass count
		for ( int i = 0; i < SAH_PARTS; ++i )
		{
			uint64_t l = 0, r = 0;
			for ( uint64_t j = 0; j < n; ++j )
			{
				float a, b;
				get( j, a, b );
				if ( a < seps[i] ) ++l;
				if ( b > seps[i] ) ++r;
			}
			L[i] = l;
			R[i] = r;
		}
	}

	float best = 0.0f;
	for ( int i = 0; i < SAH_PARTS; ++i )
	{
		const float sep = seps[i];
		float sah = ( area + ( sep - c + e ) * ( eA + eB ) ) * L[i] + ( area + ( c + e - sep ) * ( eA + eB ) ) * R[i];
		if ( i == 0 || sah < best )
		{
			best = sah;
			outSep = sep;
			cntL = L[i];
			cntR = R[i];
		}
	}
	if ( bestCost ) *bestCost = best;
	// cost of keeping all n triangles in this box, in the same units
	if ( leafCost ) *leafCost = ( area + ( mx - mn ) * ( eA + eB ) ) * n;
}

// ---------------------------------------------------------------------------
// parallel builder producing the HBM layout
// ---------------------------------------------------------------------------

struct BNode
{
	BNode *l = nullptr, *r = nullptr;
	std::vector<uint32_t> prims;     // leaves only
	std::vector<uint32_t> full;      // leaves of a pruned tree: the complete (reference) list
	float sep = 0.0f;
	int axis = 0;
	bool leaf = false;
	uint64_t nodes = 1, refs = 0, refsFull = 0;    // sub-tree totals
	int depth = 0;                   // deepest leaf below, relative to this node
};

// lo / hi: the node's cell given by the scene box and the split planes above it (the
// planes themselves, not the rounded centre / half size of `box`)
struct Cell { float lo[3], hi[3]; };
struct Task { BNode *node; std::vector<uint32_t> prims; rt_box box; int depth; Cell cell; };

// Does the triangle touch the cell grown by `pad` on every side?  Separating-axis
// test (3 cell normals, the triangle's normal, 9 edge cross products) in double
// precision; answers "yes" in every doubtful case.
bool tri_touches_cell( const rt_vector3 &q0, const rt_vector3 &q1, const rt_vector3 &q2, const Cell &c, double pad )
{
	double bc[3], bh[3], a[3][3];
	const float *q[3] = { &q0.x, &q1.x, &q2.x };
	for ( int k = 0; k < 3; ++k )
	{
		bc[k] = 0.5 * ( (double) c.lo[k] + (double) c.hi[k] );
		bh[k] = 0.5 * ( (double) c.hi[k] - (double) c.lo[k] ) + pad;
		for ( int v = 0; v < 3; ++v )
			a[v][k] = (double) q[v][k] - bc[k];
	}
	for ( int k = 0; k < 3; ++k )
	{
		// a coordinate that is not a number: keep the triangle (std::min / std::max would
		// drop a NaN that is not their first argument; the GPU builder tests all three too)
		if ( a[0][k] != a[0][k] || a[1][k] != a[1][k] || a[2][k] != a[2][k] )
			return true;
		const double mn = std::min( a[0][k], std::min( a[1][k], a[2][k] ) ), mx = std::max( a[0][k], std::max( a[1][k], a[2][k] ) );
		if ( !( mn <= bh[k] && mx >= -bh[k] ) )
			return false;
	}
	double e[3][3];
	for ( int k = 0; k < 3; ++k )
	{
		e[0][k] = a[1][k] - a[0][k];
		e[1][k] = a[2][k] - a[1][k];
		e[2][k] = a[0][k] - a[2][k];
	}
	// triangle plane
	{
		const double n[3] = { e[0][1] * -e[2][2] - e[0][2] * -e[2][1], e[0][2] * -e[2][0] - e[0][0] * -e[2][2], e[0][0] * -e[2][1] - e[0][1] * -e[2][0] };
		const double d = n[0] * a[0][0] + n[1] * a[0][1] + n[2] * a[0][2];
		const double r = std::fabs( n[0] ) * bh[0] + std::fabs( n[1] ) * bh[1] + std::fabs( n[2] ) * bh[2];
		if ( std::fabs( d ) > r )
			return false;
	}
	// unit axis k x edge f
	for ( int f = 0; f < 3; ++f )
		for ( int k = 0; k < 3; ++k )
		{
			const int k1 = ( k + 1 ) % 3, k2 = ( k + 2 ) % 3;
			// u_k x e = (component k1: -e[k2], component k2: e[k1])
			const double ax1 = -e[f][k2], ax2 = e[f][k1];
			const double p0 = ax1 * a[0][k1] + ax2 * a[0][k2], p1 = ax1 * a[1][k1] + ax2 * a[1][k2], p2 = ax1 * a[2][k1] + ax2 * a[2][k2];
			const double r = std::fabs( ax1 ) * bh[k1] + std::fabs( ax2 ) * bh[k2];
			if ( std::min( p0, std::min( p1, p2 ) ) > r || std::max( p0, std::max( p1, p2 ) ) < -r )
				return false;
		}
	return true;
}

struct Pool
{
	std::mutex mu;
	std::condition_variable cv;
	std::deque<Task> q;
	size_t pending = 0;
	bool failed = false;
};

struct Builder
{
	const float *tmin[3], *tmax[3];
	int maxDepth, maxPrims;
	float sahStop = 0.0f;      // > 0: leaf when bestCost >= sahStop * leafCost (not in the reference)
	int allAxes = 0;           // 1: try all three axes, keep the cheapest (not in the reference)
	// > 0: a leaf keeps only the triangles that touch its cell grown by pruneLeaves x the scene size
	// (not in the reference, whose leaves hold every triangle whose bounding box overlaps the cell)
	float pruneLeaves = 0.0f;
	double prunePad = 0.0;
	const rt_vertex *verts = nullptr;
	const rt_triangle *tris = nullptr;
	std::atomic<uint64_t> dropped{ 0 }, lostInPlane{ 0 };

	void make_leaf( BNode *node, std::vector<uint32_t> &prims, const Cell &cell )
	{
		node->leaf = true;
		if ( pruneLeaves > 0.0f )
		{
			node->full = prims;
			size_t k = 0;
			for ( size_t i = 0; i < prims.size(); ++i )
			{
				const rt_triangle &t = tris[prims[i]];
				if ( tri_touches_cell( verts[t.pV0].pos, verts[t.pV1].pos, verts[t.pV2].pos, cell, prunePad ) )
					prims[k++] = prims[i];
			}
			dropped.fetch_add( prims.size() - k, std::memory_order_relaxed );
			prims.resize( k );
		}
		node->prims.swap( prims );
	}
	Pool pool;
	size_t spawnThreshold = 2048;
	// Running totals: triangles that cannot be separated double their references
	// with every level, so a deep limit can ask for a tree far beyond the 30-bit
	// links (and beyond memory).  Once a total passes the limit every further
	// node becomes a leaf at once and the build is reported as failed.
	std::atomic<uint64_t> madeNodes{ 1 }, leafRefs{ 0 };
	std::atomic<bool> tooBig{ false };
	static constexpr uint64_t LIMIT = 1ull << 30;
	// ... and beyond the machine: a node of the tree under construction takes about 96 bytes, a
	// reference 4 (twice that while it is being partitioned).  A quarter of the physical memory,
	// at most 32 GiB, is what a build may take.
	uint64_t memBudget = mem_budget();
	static uint64_t mem_budget()
	{
		const long pages = sysconf( _SC_PHYS_PAGES ), sz = sysconf( _SC_PAGE_SIZE );
		uint64_t b = pages > 0 && sz > 0 ? (uint64_t) pages * (uint64_t) sz / 4u : 8ull << 30;
		return std::min<uint64_t>( b, 32ull << 30 );
	}
	void account()
	{
		if ( madeNodes.load( std::memory_order_relaxed ) * 96u + leafRefs.load( std::memory_order_relaxed ) * 8u > memBudget )
			tooBig.store( true, std::memory_order_relaxed );
	}

	void split( BNode *node, std::vector<uint32_t> &prims, const rt_box &box, int depth, bool mayspawn, const Cell &cell )
	{
		const uint64_t n = prims.size();
		if ( tooBig.load( std::memory_order_relaxed ) )
		{
			node->leaf = true;
			return;
		}
		if ( n <= (uint64_t) maxPrims || depth == maxDepth )
		{
			make_leaf( node, prims, cell );
			if ( leafRefs.fetch_add( n, std::memory_order_relaxed ) + n >= LIMIT )
				tooBig.store( true, std::memory_order_relaxed );
			return;
		}
		int ax = split_axis( box );
		uint64_t cl = 0, cr = 0;
		float sep = 0.0f, best = 0.0f, leafCost = 0.0f;
		const uint32_t *pp = prims.data();
		{
			const float *mnA = tmin[ax], *mxA = tmax[ax];
			sah_sweep( n, ax, box, [&]( uint64_t j, float &a, float &b ) { a = mnA[pp[j]]; b = mxA[pp[j]]; }, cl, cr, sep, &best, &leafCost );
		}
		if ( allAxes )
			for ( int a2 = 0; a2 < 3; ++a2 )
			{
				if ( a2 == split_axis( box ) )
					continue;
				const float *mnB = tmin[a2], *mxB = tmax[a2];
				uint64_t l2 = 0, r2 = 0;
				float s2 = 0.0f, b2 = 0.0f;
				sah_sweep( n, a2, box, [&]( uint64_t j, float &a, float &b ) { a = mnB[pp[j]]; b = mxB[pp[j]]; }, l2, r2, s2, &b2 );
				if ( b2 < best )
				{
					best = b2; ax = a2; cl = l2; cr = r2; sep = s2;
				}
			}
		if ( sahStop > 0.0f && best >= sahStop * leafCost )
		{
			make_leaf( node, prims, cell );
			return;
		}
		const float *mnA = tmin[ax], *mxA = tmax[ax];

		std::vector<uint32_t> pl, pr;
		pl.reserve( cl );
		pr.reserve( cr );
		uint64_t neither = 0;
		for ( uint64_t i = 0; i < n; ++i )        // stable partition (:971-1029)
		{
			const uint32_t p = pp[i];
			const bool toL = mnA[p] < sep, toR = mxA[p] > sep;
			if ( toL ) pl.push_back( p );
			if ( toR ) pr.push_back( p );
			neither += !toL && !toR;
		}
		if ( neither )                            // (a triangle lying in the split plane goes to neither child)
			lostInPlane.fetch_add( neither, std::memory_order_relaxed );
		std::vector<uint32_t>().swap( prims );

		rt_box bl, br;
		child_boxes( box, ax, sep, bl, br );
		Cell cl2 = cell, cr2 = cell;
		cl2.hi[ax] = sep;
		cr2.lo[ax] = sep;
		node->sep = sep;
		node->axis = ax;
		node->l = new BNode();
		node->r = new BNode();
		if ( madeNodes.fetch_add( 2, std::memory_order_relaxed ) + 2 >= LIMIT )
			tooBig.store( true, std::memory_order_relaxed );
		account();

		if ( mayspawn && pr.size() >= spawnThreshold && pl.size() >= spawnThreshold )
		{
			// hand the right sub-tree to another worker, keep the left one
			{
				std::lock_guard<std::mutex> g( pool.mu );
				pool.q.push_back( Task{ node->r, std::move( pr ), br, depth + 1, cr2 } );
				++pool.pending;
			}
			pool.cv.notify_one();
			split( node->l, pl, bl, depth + 1, true, cl2 );
		}
		else
		{
			split( node->l, pl, bl, depth + 1, mayspawn, cl2 );
			split( node->r, pr, br, depth + 1, mayspawn, cr2 );
		}
	}

	void worker()
	{
		for ( ;; )
		{
			Task t;
			{
				std::unique_lock<std::mutex> g( pool.mu );
				pool.cv.wait( g, [&] { return !pool.q.empty() || pool.pending == 0; } );
				if ( pool.q.empty() )
					return;
				t = std::move( pool.q.front() );
				pool.q.pop_front();
			}
			split( t.node, t.prims, t.box, t.depth, true, t.cell );
			{
				std::lock_guard<std::mutex> g( pool.mu );
				--pool.pending;
				if ( pool.pending == 0 )
					pool.cv.notify_all();
			}
		}
	}
};

void totals( BNode *n )
{
	if ( n->leaf )
	{
		n->nodes = 1;
		n->refs = n->prims.size();
		n->refsFull = n->full.size();
		n->depth = 0;
		return;
	}
	totals( n->l );
	totals( n->r );
	n->nodes = 1 + n->l->nodes + n->r->nodes;
	n->refs = n->l->refs + n->r->refs;
	n->refsFull = n->l->refsFull + n->r->refsFull;
	n->depth = 1 + std::max( n->l->depth, n->r->depth );
}

void release_tree( BNode *n )
{
	if ( !n )
		return;
	if ( !n->leaf )
	{
		release_tree( n->l );
		release_tree( n->r );
	}
	delete n;
}

// pre-order packing, left child right behind its parent (:860-912).  leafFull (pruned trees): the complete
// lists go behind all the pruned ones (fullRef counts from `fullBase`), in the same leaf order.
void pack( BNode *n, uint64_t pos, uint64_t ref, uint64_t fullRef, rt_hip_kdnode *nodes, uint32_t *order, rt_hip_leaf_full *leafFull )
{
	if ( n->leaf )
	{
		nodes[pos].a = (uint32_t) ref;
		nodes[pos].b = ( (uint32_t) n->prims.size() << 2 ) | 3u;
		if ( !n->prims.empty() )
			memcpy( order + ref, n->prims.data(), n->prims.size() * sizeof( uint32_t ) );
		if ( leafFull )
		{
			leafFull[pos].first = (uint32_t) fullRef;
			leafFull[pos].count = (uint32_t) n->full.size();
			if ( !n->full.empty() )
				memcpy( order + fullRef, n->full.data(), n->full.size() * sizeof( uint32_t ) );
		}
		delete n;
		return;
	}
	uint64_t right = pos + 1 + n->l->nodes;
	uint64_t rref = ref + n->l->refs, rfull = fullRef + n->l->refsFull;
	memcpy( &nodes[pos].a, &n->sep, 4 );
	nodes[pos].b = ( (uint32_t) right << 2 ) | (uint32_t) n->axis;
	if ( leafFull )
		leafFull[pos].first = leafFull[pos].count = 0u;
	BNode *l = n->l, *r = n->r;
	delete n;
	pack( l, pos + 1, ref, fullRef, nodes, order, leafFull );
	pack( r, right, rref, rfull, nodes, order, leafFull );
}

void scene_box( const rt_vertex *verts, uint64_t n, rt_box &bbox )
{
	// :1114-1137: min / max over all vertices, padded by 1, centre + half size
	float mnx = 0, mny = 0, mnz = 0, mxx = 0, mxy = 0, mxz = 0;
	for ( uint64_t i = 0; i < n; ++i )
	{
		const rt_vector3 &p = verts[i].pos;
		mnx = ( mnx > p.x || i == 0 ) ? p.x : mnx;
		mny = ( mny > p.y || i == 0 ) ? p.y : mny;
		mnz = ( mnz > p.z || i == 0 ) ? p.z : mnz;
		mxx = ( mxx < p.x || i == 0 ) ? p.x : mxx;
		mxy = ( mxy < p.y || i == 0 ) ? p.y : mxy;
		mxz = ( mxz < p.z || i == 0 ) ? p.z : mxz;
	}
	mnx -= 1.0f; mny -= 1.0f; mnz -= 1.0f;
	mxx += 1.0f; mxy += 1.0f; mxz += 1.0f;
	memset( &bbox, 0, sizeof( bbox ) );
	bbox.center.x = ( mnx + mxx ) * 0.5f;
	bbox.center.y = ( mny + mxy ) * 0.5f;
	bbox.center.z = ( mnz + mxz ) * 0.5f;
	bbox.extents.x = ( mxx - mnx ) * 0.5f;
	bbox.extents.y = ( mxy - mny ) * 0.5f;
	bbox.extents.z = ( mxz - mnz ) * 0.5f;
}

void fill_records( const rt_vertex *verts, const rt_triangle *tris, uint64_t nTris, const uint32_t *order,
	uint64_t nRecs, rt_hip_trirec *recs, rt_hip_trishade *shade, int threads )
{
	auto recRange = [&]( uint64_t b, uint64_t e ) {
		for ( uint64_t i = b; i < e; ++i )
		{
			const rt_triangle &t = tris[order[i]];
			const rt_vector3 &p0 = verts[t.pV0].pos, &p1 = verts[t.pV1].pos, &p2 = verts[t.pV2].pos;
			rt_hip_trirec &r = recs[i];
			r.nv0x = -p0.x; r.nv0y = -p0.y; r.nv0z = -p0.z;
			r.e1x = p1.x - p0.x; r.e1y = p1.y - p0.y; r.e1z = p1.z - p0.z;
			r.e2x = p2.x - p0.x; r.e2y = p2.y - p0.y; r.e2z = p2.z - p0.z;
			r.tri = order[i];
			r.mat = (uint32_t) t.mat;
			r._pad = 0;
		}
	};
	auto shadeRange = [&]( uint64_t b, uint64_t e ) {
		for ( uint64_t i = b; i < e; ++i )
		{
			const rt_triangle &t = tris[i];
			const rt_vector3 &n0 = verts[t.pV0].norm, &n1 = verts[t.pV1].norm, &n2 = verts[t.pV2].norm;
			rt_hip_trishade &s = shade[i];
			s.n0[0] = n0.x; s.n0[1] = n0.y; s.n0[2] = n0.z;
			s.n1[0] = n1.x; s.n1[1] = n1.y; s.n1[2] = n1.z;
			s.n2[0] = n2.x; s.n2[1] = n2.y; s.n2[2] = n2.z;
			s.mat = (uint32_t) t.mat;
			s._pad[0] = s._pad[1] = 0;
		}
	};
	if ( threads <= 1 || nRecs + nTris < 65536 )
	{
		recRange( 0, nRecs );
		shadeRange( 0, nTris );
		return;
	}
	std::vector<std::thread> th;
	for ( int k = 0; k < threads; ++k )
		th.emplace_back( [&, k] {
			recRange( nRecs * k / threads, nRecs * ( k + 1 ) / threads );
			shadeRange( nTris * k / threads, nTris * ( k + 1 ) / threads );
		} );
	for ( auto &t : th ) t.join();
}

int pick_threads( int threads )
{
	if ( threads > 0 )
		return std::min( threads, 64 );
	long n = sysconf( _SC_NPROCESSORS_ONLN );
	if ( const char *e = getenv( "OMP_NUM_THREADS" ) )
	{
		int v = atoi( e );
		if ( v > 0 && v < n ) n = v;
	}
	if ( n < 1 ) n = 1;
	return (int) std::min<long>( n, 16 );
}

} // namespace

extern "C" int rt_kdtree_build_flat( const rt_vertex *verts, const rt_triangle *tris, uint64_t vertexCount,
	uint64_t trianglesCount, int maxDepth, int maxPrims, int threads, rt_kdtree_flat *out )
{
	rt_kdtree_build_opts o;
	memset( &o, 0, sizeof( o ) );
	o.maxDepth = maxDepth;
	o.maxPrims = maxPrims;
	o.threads = threads;
	return rt_kdtree_build_flat_ex( verts, tris, vertexCount, trianglesCount, &o, out );
}

extern "C" int rt_kdtree_build_flat_ex( const rt_vertex *verts, const rt_triangle *tris, uint64_t vertexCount,
	uint64_t trianglesCount, const rt_kdtree_build_opts *opts, rt_kdtree_flat *out )
{
	if ( !opts )
		return 1;
	const int maxDepth = opts->maxDepth, maxPrims = opts->maxPrims;
	int threads = opts->threads;
	if ( !out || ( vertexCount && !verts ) || ( trianglesCount && !tris ) )
		return 1;
	if ( trianglesCount >= ( 1ull << 30 ) || maxDepth < 0 || maxDepth > RT_HIP_MAX_TREE_DEPTH || maxPrims < 1 )
		return 2;
	for ( uint64_t i = 0; i < trianglesCount; ++i )
		if ( tris[i].pV0 >= vertexCount || tris[i].pV1 >= vertexCount || tris[i].pV2 >= vertexCount )
			return 3;
	memset( out, 0, sizeof( *out ) );
	threads = pick_threads( threads );
	scene_box( verts, vertexCount, out->bbox );

	const uint64_t nT = trianglesCount;
	std::vector<float> ext( 6 * ( nT ? nT : 1 ) );
	{
		auto range = [&]( uint64_t b, uint64_t e ) {
			for ( uint64_t i = b; i < e; ++i )
			{
				const rt_vector3 &p0 = verts[tris[i].pV0].pos, &p1 = verts[tris[i].pV1].pos, &p2 = verts[tris[i].pV2].pos;
				tri_extent( p0.x, p1.x, p2.x, ext[i], ext[3 * nT + i] );
				tri_extent( p0.y, p1.y, p2.y, ext[nT + i], ext[4 * nT + i] );
				tri_extent( p0.z, p1.z, p2.z, ext[2 * nT + i], ext[5 * nT + i] );
			}
		};
		if ( threads > 1 && nT >= 65536 )
		{
			std::vector<std::thread> th;
			for ( int k = 0; k < threads; ++k )
				th.emplace_back( range, nT * k / threads, nT * ( k + 1 ) / threads );
			for ( auto &t : th ) t.join();
		}
		else
			range( 0, nT );
	}

	Builder b;
	for ( int a = 0; a < 3; ++a )
	{
		b.tmin[a] = ext.data() + a * nT;
		b.tmax[a] = ext.data() + ( 3 + a ) * nT;
	}
	b.maxDepth = maxDepth;
	b.maxPrims = maxPrims;
	b.sahStop = opts->sahStop;
	b.allAxes = opts->allAxes;
	b.pruneLeaves = opts->pruneLeaves > 0.0f ? opts->pruneLeaves : 0.0f;
	b.verts = verts;
	b.tris = tris;
	Cell rootCell;
	{
		const float c[3] = { out->bbox.center.x, out->bbox.center.y, out->bbox.center.z };
		const float e[3] = { out->bbox.extents.x, out->bbox.extents.y, out->bbox.extents.z };
		double size = 0.0;
		for ( int k = 0; k < 3; ++k )
		{
			rootCell.lo[k] = c[k] - e[k];
			rootCell.hi[k] = c[k] + e[k];
			size = std::max( size, std::fabs( (double) c[k] ) + std::fabs( (double) e[k] ) );
		}
		b.prunePad = (double) b.pruneLeaves * size;
	}

	BNode *root = new BNode();
	{
		std::vector<uint32_t> all( nT );
		for ( uint64_t i = 0; i < nT; ++i ) all[i] = (uint32_t) i;
		b.pool.q.push_back( Task{ root, std::move( all ), out->bbox, 0, rootCell } );
		b.pool.pending = 1;
	}
	if ( threads > 1 && nT >= 4 * b.spawnThreshold )
	{
		std::vector<std::thread> th;
		for ( int k = 0; k < threads; ++k )
			th.emplace_back( [&] { b.worker(); } );
		for ( auto &t : th ) t.join();
	}
	else
		b.worker();

	// Pruning relies on every triangle being listed by every leaf it reaches.  The reference's
	// partition drops a triangle that lies in the split plane from both children, and what such a
	// tree then returns depends on the leaves that list the triangle without containing it (a hit
	// in front of a leaf's interval is accepted).  Such a tree keeps the reference's leaf lists.
	if ( b.pruneLeaves > 0.0f && b.lostInPlane.load() != 0 )
	{
		release_tree( root );
		rt_kdtree_build_opts plain = *opts;
		plain.pruneLeaves = 0.0f;
		return rt_kdtree_build_flat_ex( verts, tris, vertexCount, trianglesCount, &plain, out );
	}
	totals( root );
	// (a pruned tree carries the complete lists behind the pruned ones - unless nothing was pruned)
	const bool twoLists = b.pruneLeaves > 0.0f && b.dropped.load() != 0;
	if ( b.tooBig.load() || root->nodes >= ( 1ull << 30 ) || root->refs + ( twoLists ? root->refsFull : 0 ) >= ( 1ull << 30 ) )
	{
		release_tree( root );
		return 4;          // 30-bit links of the 8-byte node, or the memory a build may take
	}
	out->nNodes = root->nodes;
	out->nRecs = root->refs;
	out->nRecsFull = twoLists ? root->refsFull : 0;
	out->nTris = nT;
	out->depth = root->depth;
	const uint64_t nAll = out->nRecs + out->nRecsFull;
	out->nodes = (rt_hip_kdnode *) malloc( sizeof( rt_hip_kdnode ) * out->nNodes );
	out->recs = (rt_hip_trirec *) malloc( sizeof( rt_hip_trirec ) * ( nAll ? nAll : 1 ) );
	out->shade = (rt_hip_trishade *) malloc( sizeof( rt_hip_trishade ) * ( nT ? nT : 1 ) );
	out->leafFull = twoLists ? (rt_hip_leaf_full *) malloc( sizeof( rt_hip_leaf_full ) * out->nNodes ) : nullptr;
	std::vector<uint32_t> order( nAll ? nAll : 1 );
	pack( root, 0, 0, out->nRecs, out->nodes, order.data(), out->leafFull );
	fill_records( verts, tris, nT, order.data(), nAll, out->recs, out->shade, threads );
	return 0;
}

extern "C" void rt_kdtree_flat_release( rt_kdtree_flat *t )
{
	if ( !t )
		return;
	free( t->nodes );
	free( t->recs );
	free( t->shade );
	free( t->leafFull );
	memset( t, 0, sizeof( *t ) );
}

extern "C" int rt_kdtree_flat_expand( const rt_kdtree_flat *t, rt_cl_kdtree_node **nodes, uint64_t *nodesCount,
	rt_ulong **primsIdx, uint64_t *primsIdxCount )
{
	if ( !t || !nodes || !nodesCount || !primsIdx || !primsIdxCount )
		return 1;
	rt_cl_kdtree_node *n = (rt_cl_kdtree_node *) calloc( t->nNodes ? t->nNodes : 1, sizeof( rt_cl_kdtree_node ) );
	rt_ulong *idx = (rt_ulong *) calloc( t->nRecs ? t->nRecs : 1, sizeof( rt_ulong ) );
	for ( uint64_t i = 0; i < t->nNodes; ++i )
	{
		const rt_hip_kdnode &s = t->nodes[i];
		if ( ( s.b & 3u ) == 3u )
		{
			n[i].isLast = 1;
			n[i].primsCount = s.b >> 2;
			n[i].prims = s.a;
		}
		else
		{
			memcpy( &n[i].sep, &s.a, 4 );
			n[i].axis = (RT_AXIS) ( s.b & 3u );
			n[i].leftNode = i + 1;
			n[i].rightNode = s.b >> 2;
		}
	}
	for ( uint64_t i = 0; i < t->nRecs; ++i )
		idx[i] = t->recs[i].tri;
	*nodes = n;
	*nodesCount = t->nNodes;
	*primsIdx = idx;
	*primsIdxCount = t->nRecs;
	return 0;
}

extern "C" int rt_kdtree_flatten( const rt_cl_kdtree_node *nodes, uint64_t nodesCount, const rt_ulong *primsIdx,
	uint64_t primsIdxCount, const rt_vertex *verts, const rt_triangle *tris, uint64_t vertexCount,
	uint64_t trianglesCount, const rt_box *bbox, rt_kdtree_flat *out )
{
	if ( !nodes || !nodesCount || !out || !bbox || ( primsIdxCount && !primsIdx ) )
		return 1;
	if ( nodesCount >= ( 1ull << 30 ) || primsIdxCount >= ( 1ull << 30 ) || trianglesCount >= ( 1ull << 30 ) )
		return 2;
	for ( uint64_t i = 0; i < trianglesCount; ++i )
		if ( tris[i].pV0 >= vertexCount || tris[i].pV1 >= vertexCount || tris[i].pV2 >= vertexCount )
			return 3;
	memset( out, 0, sizeof( *out ) );
	out->bbox = *bbox;
	out->nNodes = nodesCount;
	out->nRecs = primsIdxCount;
	out->nTris = trianglesCount;
	out->nodes = (rt_hip_kdnode *) malloc( sizeof( rt_hip_kdnode ) * nodesCount );
	out->recs = (rt_hip_trirec *) malloc( sizeof( rt_hip_trirec ) * ( primsIdxCount ? primsIdxCount : 1 ) );
	out->shade = (rt_hip_trishade *) malloc( sizeof( rt_hip_trishade ) * ( trianglesCount ? trianglesCount : 1 ) );
	std::vector<uint32_t> order( primsIdxCount ? primsIdxCount : 1 );
	std::vector<int> depth( nodesCount, 0 );
	uint64_t nextRef = 0;
	int deepest = 0;
	for ( uint64_t i = 0; i < nodesCount; ++i )
	{
		const rt_cl_kdtree_node &s = nodes[i];
		deepest = std::max( deepest, depth[i] );
		if ( s.isLast )
		{
			// leaf lists must follow each other in node order, as the reference packs them
			if ( s.prims != nextRef || s.prims + s.primsCount > primsIdxCount )
			{
				rt_kdtree_flat_release( out );
				return 5;
			}
			out->nodes[i].a = (uint32_t) s.prims;
			out->nodes[i].b = ( (uint32_t) s.primsCount << 2 ) | 3u;
			nextRef += s.primsCount;
		}
		else
		{
			if ( s.leftNode != i + 1 || s.rightNode <= i + 1 || s.rightNode >= nodesCount || (unsigned) s.axis > 2u )
			{
				rt_kdtree_flat_release( out );
				return 6;
			}
			memcpy( &out->nodes[i].a, &s.sep, 4 );
			out->nodes[i].b = ( (uint32_t) s.rightNode << 2 ) | (uint32_t) s.axis;
			depth[i + 1] = depth[s.rightNode] = depth[i] + 1;
		}
	}
	for ( uint64_t i = 0; i < primsIdxCount; ++i )
	{
		if ( primsIdx[i] >= trianglesCount )
		{
			rt_kdtree_flat_release( out );
			return 7;
		}
		order[i] = (uint32_t) primsIdx[i];
	}
	out->depth = deepest;
	fill_records( verts, tris, trianglesCount, order.data(), primsIdxCount, out->recs, out->shade, 1 );
	return 0;
}

// ---------------------------------------------------------------------------
// reference-signature builder entry points (pointer tree, malloc'ed lists)
// ---------------------------------------------------------------------------

static inline float axis_of( const rt_vector3 &p, int ax ) { return ax == 0 ? p.x : ax == 1 ? p.y : p.z; }

extern "C" void rt_kdtree_compute_sah( rt_ulong *memTp, rt_triangle *memTr, rt_vertex *memVer, rt_ulong primsCount,
	RT_AXIS axis, rt_box *pBox, rt_ulong *primsCountL, rt_ulong *primsCountR, float *pSep )
{
	if ( !memTp || !memTr || !memVer || !pBox || !primsCountL || !primsCountR || !pSep )
		exit( -1 );
	uint64_t l = 0, r = 0;
	float sep = 0.0f;
	sah_sweep( primsCount, (int) axis, *pBox, [&]( uint64_t j, float &a, float &b ) {
		const rt_triangle &t = memTr[memTp[j]];
		tri_extent( axis_of( memVer[t.pV0].pos, axis ), axis_of( memVer[t.pV1].pos, axis ),
			axis_of( memVer[t.pV2].pos, axis ), a, b );
	}, l, r, sep );
	*primsCountL = l;
	*primsCountR = r;
	*pSep = sep;
}

extern "C" rt_kdtree_count_info rt_kdtree_make_childs( rt_vertex *memVer, rt_triangle *memTr, rt_kdtree_node *pNode,
	rt_box *bbox, int depth )
{
	if ( !memVer || !memTr || !pNode || !bbox )
		exit( -1 );
	rt_kdtree_count_info info;
	if ( pNode->primsCount <= MAX_PRIMS_IN_NODE || depth == MAX_DEPTH )
	{
		pNode->isLast = 1;
		pNode->leftNode = pNode->rightNode = NULL;
		info.primsCount = pNode->primsCount;
		info.nodesCount = 1;
		return info;
	}
	pNode->isLast = 0;
	const int ax = split_axis( *bbox );
	pNode->axis = (RT_AXIS) ax;
	pNode->leftNode = (rt_kdtree_node *) malloc( sizeof( rt_kdtree_node ) );
	pNode->rightNode = (rt_kdtree_node *) malloc( sizeof( rt_kdtree_node ) );
	rt_kdtree_compute_sah( pNode->prims, memTr, memVer, pNode->primsCount, pNode->axis, bbox,
		&pNode->leftNode->primsCount, &pNode->rightNode->primsCount, &pNode->sep );
	pNode->leftNode->prims = (rt_ulong *) malloc( sizeof( rt_ulong ) * ( pNode->leftNode->primsCount + 1 ) );
	pNode->rightNode->prims = (rt_ulong *) malloc( sizeof( rt_ulong ) * ( pNode->rightNode->primsCount + 1 ) );
	rt_ulong j = 0, k = 0;
	for ( rt_ulong i = 0; i < pNode->primsCount; ++i )
	{
		const rt_triangle &t = memTr[pNode->prims[i]];
		float mn, mx;
		tri_extent( axis_of( memVer[t.pV0].pos, ax ), axis_of( memVer[t.pV1].pos, ax ), axis_of( memVer[t.pV2].pos, ax ), mn, mx );
		if ( mn < pNode->sep ) pNode->leftNode->prims[j++] = pNode->prims[i];
		if ( mx > pNode->sep ) pNode->rightNode->prims[k++] = pNode->prims[i];
	}
	free( pNode->prims );
	pNode->prims = NULL;
	rt_box bl, br;
	child_boxes( *bbox, ax, pNode->sep, bl, br );
	rt_kdtree_count_info a = rt_kdtree_make_childs( memVer, memTr, pNode->leftNode, &bl, depth + 1 );
	rt_kdtree_count_info b = rt_kdtree_make_childs( memVer, memTr, pNode->rightNode, &br, depth + 1 );
	info.primsCount = a.primsCount + b.primsCount;
	info.nodesCount = a.nodesCount + b.nodesCount + 1;
	return info;
}

// Flattens and frees the pointer tree.  Like the reference (:860-912) the
// root and every inner node are freed; unlike it, leaf nodes are freed too.
extern "C" rt_kdtree_count_info rt_kdtree_pack_to_buffer( rt_cl_kdtree_node *pNodeBuf, rt_ulong *pPrimsIdxBuf,
	rt_kdtree_node *pNode, rt_ulong writePos, rt_ulong primsWritePos )
{
	if ( !pNodeBuf || !pNode )
		exit( -1 );
	rt_kdtree_count_info info;
	pNodeBuf[writePos].isLast = pNode->isLast;
	if ( pNode->isLast )
	{
		info.nodesCount = 1;
		info.primsCount = pNode->primsCount;
		pNodeBuf[writePos].primsCount = pNode->primsCount;
		pNodeBuf[writePos].prims = primsWritePos;
		if ( pNode->primsCount && pPrimsIdxBuf )
			memcpy( pPrimsIdxBuf + primsWritePos, pNode->prims, sizeof( rt_ulong ) * pNode->primsCount );
		free( pNode->prims );
		if ( writePos != 0 )
			free( pNode );
		return info;
	}
	rt_kdtree_count_info l = rt_kdtree_pack_to_buffer( pNodeBuf, pPrimsIdxBuf, pNode->leftNode, writePos + 1, primsWritePos );
	rt_kdtree_count_info r = rt_kdtree_pack_to_buffer( pNodeBuf, pPrimsIdxBuf, pNode->rightNode,
		writePos + l.nodesCount + 1, primsWritePos + l.primsCount );
	pNodeBuf[writePos].sep = pNode->sep;
	pNodeBuf[writePos].axis = pNode->axis;
	pNodeBuf[writePos].leftNode = writePos + 1;
	pNodeBuf[writePos].rightNode = writePos + l.nodesCount + 1;
	free( pNode );
	info.nodesCount = l.nodesCount + r.nodesCount + 1;
	info.primsCount = l.primsCount + r.primsCount;
	return info;
}
