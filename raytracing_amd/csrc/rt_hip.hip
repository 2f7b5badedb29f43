// HIP layer + CDNA4 kernels of the ray tracer (gfx950 only).
//
// One frame is a short sequence of persistent wave64 kernels over explicit
// work lists in HBM instead of one per-pixel megakernel:
//
//   for slot k = 0 .. STACK_SIZE-1            (slot = index in the pixel's ray list)
//     k_closest<k>  ray list k  -> closest hit (spheres brute force + kd tree);
//                   misses finish their slot, hits are COMPACTED into a hit list
//                   and shaded as far as the next slot needs it: hit point,
//                   normal, child rays appended (wave-aggregated atomics) to the
//                   ray lists of the slots they occupy
//     k_shadow<k>   hit list k x lights -> one shadow traversal per item (lean
//                   register state: only what the traversal needs)
//     k_light<k>    hit list k + shadow sums -> Phong terms, slot colour
//   k_fold          per pixel: fold child colours into parents back to front,
//                   clamp, truncate to 8-bit BGRA
//
// k_closest<k> -> k_closest<k+1> is the only dependency chain; the shadow and
// lighting passes of a slot hang off it.  The deep slots hold few but very
// long rays (a few busy waves, the rest of the chip idle), so the chain runs
// on the caller's stream and the shadow / lighting passes on two auxiliary
// streams that fill the idle part of the chip.
//
// This is the reference's iterative ray tree (CL/rt_funcs_primitives.cl:584-728)
// turned inside out: the reference walks the <= 6 slots of ONE pixel in a
// private array; here every launch processes slot k of ALL pixels, so lanes
// are always packed with live rays, and the slot a child lands in is the same
// as in the reference because slots of a pixel are still filled in slot order.
//
// Waves pull 64-entry chunks from a device-side counter until the list is
// empty (persistent workers); the kd traversal stack lives in LDS
// (RT_LDS_STACK entries per lane, lane-interleaved, conflict free) and spills
// to a per-lane strip of HBM beyond that.
//
// Arithmetic is fp32 without contraction, in the operand order of the
// reference's OpenCL source, so hit ids and barycentrics agree bit for bit
// with the CPU restatement used by the tests.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "rt_hip.h"
#include "rt_kdtree.h"

// One wave per workgroup: the persistent kernels of the frames in flight overlap, and a
// workgroup can only start where all its waves find room at once and gives its LDS back only
// when its last wave has ended.  Step time of the benchmark frame, three frames in flight,
// 8 waves per CU and launch (RT_OCC_CAP): 5.67 ms with four waves per workgroup, 5.52 ms with
// two, 5.38 ms with one (nothing in a frame kernel needs more than a wave: no barriers, LDS
// per wave).
#ifndef RT_BLOCK
#define RT_BLOCK      64
#endif
#ifndef RT_LDS_STACK
#define RT_LDS_STACK  12     // traversal stack entries per lane kept in LDS
#endif
// Resident waves per SIMD the traversal kernels are compiled for: 5 (96 vector registers), without scratch
// memory.  What made room for the fifth wave was not the traversal itself but what sat beside it for the whole
// launch: values that are the same for every ray (scene-box planes, camera origin and matrices, pixel scale,
// division constants, the address of the lane's stack-overflow strip) are hoisted out of the chunk loop by the
// compiler, and uniform FLOAT values live in vector registers on this ISA; the scalar registers were oversubscribed
// by kernel arguments, seven reservation cursors and the list addresses derived from them, and scalar overflow is
// kept in vector registers as well.  Now the host derives the per-frame constants (derive_scene), the camera
// matrices are read from device memory where a chunk's rays are made, the cursors live in LDS, child rays are
// appended in a loop over the slots that occur instead of an unrolled one over all, and wave-uniform values are
// declared as such (wave_bcast): 119 -> 96 registers for k_closest, 105 -> 89 for k_shadow, step time of the
// benchmark frame -5.5 % (measured with a 96-register build that still spilled 80 B per lane outside its loops).
#ifndef RT_WAVES_CLOSEST
#define RT_WAVES_CLOSEST 5
#endif
#ifndef RT_WAVES_CLOSEST_STATS
#define RT_WAVES_CLOSEST_STATS 3     // the counting build needs more registers
#endif
#ifndef RT_WAVES_CLOSEST_BIG
#define RT_WAVES_CLOSEST_BIG RT_WAVES_CLOSEST
#endif
#ifndef RT_BIG_FRAME
#define RT_BIG_FRAME 0xffffffffu
#endif
#ifndef RT_WAVES_SHADOW
#define RT_WAVES_SHADOW  5   // the same for the shadow kernel
#endif
#ifndef RT_WAVES_SHADOW_STATS
#define RT_WAVES_SHADOW_STATS 4
#endif
#define RT_SPILL_CAP  (RT_HIP_MAX_TREE_DEPTH - RT_LDS_STACK)
#define RT_MAX_SLOTS  RT_STACK_SIZE_MAX
#define RT_TIMED_FRAMES 32
#ifndef RT_AUX_OCC
#define RT_AUX_OCC    ( 1024 / RT_BLOCK )      // development knob: a tighter limit (workgroups per CU) for a shadow pass that runs beside the closest-hit chain; the default, 16 waves, never binds (RT_PARAM_LAUNCH_WAVES does)
#endif
#ifndef RT_AUX_STREAMS
#define RT_AUX_STREAMS 2      // auxiliary streams the shadow / lighting passes are dealt to
#endif
#ifndef RT_TRAV_DYN
#define RT_TRAV_DYN   8u     // eighths of a list the traversal passes hand out dynamically (8 = all but the first round)
#endif
#ifndef RT_STREAM_DYN
#define RT_STREAM_DYN 1u     // the same for the streaming passes
#endif
#ifndef RT_RESV_BLOCK
#define RT_RESV_BLOCK 256u   // list entries a wave reserves per atomic when the list is long
#endif
// Gear change of the traversal passes (see "Cooperative walk"): a wave goes over
// to the cooperative walk once at most RT_COOP_LIVE of its lanes are still
// walking (0 = never); short lists are dealt out in chunks of as few as
// 2^RT_COOP_MIN_SHIFT rays per wave.
#ifndef RT_COOP_LIVE
#define RT_COOP_LIVE 32
#endif
#ifndef RT_COOP_MIN_SHIFT
#define RT_COOP_MIN_SHIFT 3u
#endif
// Waves per CU of one traversal pass (RT_PARAM_LAUNCH_WAVES; 16 fit).  A frame on its own is fastest
// when every pass takes all of them (6.7 ms for the benchmark frame against 7.4 ms with 8).  With
// frames in flight it is the other way round: a pass that takes every wave slot leaves none for the
// passes beside it and for the other frames, which then only get in as its waves retire - step time
// with three frames in flight and at most 4 / 6 / 8 / 10 / 12 / 16 waves per CU and pass: 5.76 / 5.46 /
// 5.38 / 5.40 / 5.50 / 5.76 ms (a 1/8 frame: 1.09 / 1.04 / 1.08 / 1.09 / 1.10 / 1.05 ms).  The default
// serves the caller that renders one frame at a time; callers that keep frames in flight ask for 8.
#define RT_MAX_LAUNCH_WAVES 20          // what fits a CU at five waves per SIMD
#ifndef RT_OCC_CAP
#define RT_OCC_CAP 16
#endif
#ifndef RT_STREAM_OCC
#define RT_STREAM_OCC ( 256 / RT_BLOCK )       // workgroups per CU of the streaming (non-traversal) passes: 4 waves (with 16, the lighting waves, 129 registers each, hold wave slots the traversal passes could use: 5.40 against 5.35 ms)
#endif

// ---------------------------------------------------------------------------
// device-side data
// ---------------------------------------------------------------------------

struct f3 { float x, y, z; };

struct DevScene
{
	const uint2  *nodes;      // rt_hip_kdnode
	const float4 *recs;       // rt_hip_trirec as 3 x float4
	// pruned leaf lists only (else null): per node, (first record, count) of the leaf's COMPLETE list, which lies
	// behind the pruned lists in `recs`; read by rays that are parallel to an axis (see axis_parallel)
	const uint2  *leafFull;
	const float4 *shade;      // rt_hip_trishade as 3 x float4
	const rt_material *mats;
	const float4 *spheres;    // (pos.xyz, rad)
	const uint32_t *sphereMat;
	const float4 *lights;     // 3 x float4: (col b,g,r,a) (pos.xyz,_) (rad,_,_,_)
	float bbC[3], bbE[3];
	float world[16], vtp[16];
	// Derived on the host when a frame is enqueued (derive_scene), with the very fp32 operations the kernels
	// would otherwise repeat: values that are the same for every ray are operands from scalar registers this
	// way; computed in the kernel they are hoisted out of its loops into vector registers that stay occupied
	// for the whole launch (17 of them).
	float bbLo[3], bbHi[3];   // bbC - bbE, bbC + bbE (the planes of CL/rt_intersection.cl:170-187)
	float xTr, yTr;           // 2 / (w - 1), 2 / (h - 1) (CL/rt_raytrace.cl:27-28)
	float camSrc[3];          // origin of every camera ray: (0,0,0) through the world matrix (CL/rt_raytrace.cl:65-67)
	// world[16] then vtp[16] again, in device memory (written by k_put_camera at the start of the frame): the
	// closest-hit pass of slot 0 reads the 18 matrix elements it needs from here when it makes a chunk's rays,
	// instead of holding them in scalar registers during the traversal
	const float *cam;
	int nSpheres, nLights;
	int w, h;                 // full image
	int localH, bandH, nParts, part;
	int stackSize;
	int alphaMonotone;        // every material opacity is >= 0: a shadow sum that reached 1 stays >= 1
};

// Every list counter sits in a cache line of its own: all waves of a launch
// hammer two of them (chunk dispenser, append position), and atomics on one
// line are executed one after the other by the L2 channel that owns it.
struct alignas( 128 ) Ctr { uint32_t v; uint32_t pad[31]; };
struct Counters
{
	Ctr rayCount[RT_MAX_SLOTS];           // entries in ray list k (k >= 1)
	Ctr headClosest[RT_MAX_SLOTS];        // chunk dispenser of k_closest
	Ctr hitCount[RT_MAX_SLOTS];           // entries in the hit list of slot k
	Ctr headShade[RT_MAX_SLOTS];          // chunk dispenser of k_light
	Ctr headShadow[RT_MAX_SLOTS];         // chunk dispenser of k_shadow
	uint32_t overflow;                    // a list append did not fit its buffer (host reports an error)
#ifdef RT_EXP_REPLAY
	Ctr rayCountAlt[RT_MAX_SLOTS];        // development build: where a replayed pass counts the child rays it throws away
#endif
#ifdef RT_PROFILE_CHUNKS
	// development build: per slot, histogram of chunk durations (10 us bins) and wave-level round counts
	uint32_t chunkHist[RT_MAX_SLOTS][64];
	unsigned long long lockRounds[RT_MAX_SLOTS], coopRounds[RT_MAX_SLOTS], coopIters[RT_MAX_SLOTS], chunksSeen[RT_MAX_SLOTS];
	unsigned long long tExpand[RT_MAX_SLOTS], tScan[RT_MAX_SLOTS], tCoop[RT_MAX_SLOTS], tChunk[RT_MAX_SLOTS];
	unsigned long long tDescend[RT_MAX_SLOTS], tLeaf[RT_MAX_SLOTS], tFinish[RT_MAX_SLOTS], tSetup[RT_MAX_SLOTS];
	uint32_t maxLock[RT_MAX_SLOTS], maxCoop[RT_MAX_SLOTS], maxIters[RT_MAX_SLOTS];
	// direction octants of the rays of a chunk: distinct ones, rays in the most frequent one, rays
	unsigned long long octDistinct[RT_MAX_SLOTS], octMajor[RT_MAX_SLOTS], octRays[RT_MAX_SLOTS];
#endif
	unsigned long long nodeVisits, triTests, spillPushes, maxStack;   // STATS builds only
	unsigned long long nodeVisitsShadow, triTestsShadow;              // share of the shadow queries
	unsigned long long nodeSlots, triSlots, nodeSlotsShadow, triSlotsShadow;   // lane-slots spent (STATS)
	unsigned long long validRays[RT_MAX_SLOTS], validHits[RT_MAX_SLOTS];      // filled by k_count_valid on request
};

struct Work
{
	Counters *ctr;
	float4 *rayA, *rayB, *rayC, *rayD;    // [(slot-1) * npix + entry]
	// hit lists, one per slot ([slot * npix + hit]): the passes of different
	// slots overlap in time
	float4 *hitA;                         // (entry, t, u, v)
	uint2  *hitB;                         // (id | sphere flag, b)
	float4 *hitN;                         // (shading normal xyz, material slot), written by k_closest
	float4 *res;                          // [slot * npix + pix] (col rgb, parent slot)
	float4 *fac;                          // [slot * npix + pix] weight of this slot in its parent
	float  *shadow;                       // [(slot * nLights + light) * npix + hit] summed shadow opacity
	uint32_t *stackPos;                   // last used slot of each pixel
	int32_t *hitId;                       // optional primary hit ids
	uint2  *spill;                        // traversal stack overflow
	uint32_t coopLive;                    // lanes still walking at which a wave goes over to the cooperative walk (0 = never)
	uint32_t coopCap;                     // stack entries per ray the cooperative walk may use (<= RT_LDS_STACK; less only in tests)
	uint32_t npix, tilesX, nChunks0, spillStride;
	uint32_t cap;                         // entries per ray list / hit list (npix + room for reservation holes)
	uint32_t exactAppends;                // 1 = lists are appended without block reservations (no holes: a list of cap = npix entries cannot overflow)
#ifdef RT_EXP_REPLAY
	// Development build (timing experiment, see enqueue_replay): the passes of ALL slots of a frame run side by side
	// over the ray lists a normal frame left behind; child rays go to a second set of lists that nobody reads.
	uint32_t replay;
	size_t altOffset;                     // entries between a ray list and its throw-away twin
#endif
};

struct Ray { f3 src, dest, inv; };

#define DEVI __device__ __forceinline__

DEVI f3 F3( float x, float y, float z ) { f3 r = { x, y, z }; return r; }
DEVI f3 add3( f3 a, f3 b ) { return F3( a.x + b.x, a.y + b.y, a.z + b.z ); }
DEVI f3 sub3( f3 a, f3 b ) { return F3( a.x - b.x, a.y - b.y, a.z - b.z ); }
DEVI f3 mul3( f3 a, float s ) { return F3( a.x * s, a.y * s, a.z * s ); }
DEVI f3 div3( f3 a, float s ) { return F3( a.x / s, a.y / s, a.z / s ); }
DEVI f3 rcp3( f3 a ) { return F3( 1.0f / a.x, 1.0f / a.y, 1.0f / a.z ); }
DEVI float dot3( f3 a, f3 b ) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEVI f3 cross3( f3 a, f3 b ) { return F3( a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x ); }
DEVI float len3( f3 a ) { return sqrtf( dot3( a, a ) ); }
DEVI f3 norm3( f3 a ) { return div3( a, len3( a ) ); }
DEVI float cl_max( float x, float y ) { return ( x < y ) ? y : x; }
// pow / exp of the reference are OpenCL built-ins whose rounding is the vendor's.
// Here: the double-precision function rounded to float - the correctly rounded
// single-precision value in all but about one case in 2^29, and therefore the
// same bits on any platform whose double functions are good to an ulp (the CPU
// checker computes them the same way).  Both are off the hot path (once per
// shaded hit and light / per refracted slot).
DEVI float cl_pow( float x, float y ) { return (float) pow( (double) x, (double) y ); }
DEVI float cl_exp( float x ) { return (float) exp( (double) x ); }
// reference CL/rt_funcs_math.cl:102-105 (NaN -> lower bound)
DEVI float clampf( float f, float b, float e ) { return ( f > b ) ? ( ( f < e ) ? f : e ) : b; }
DEVI f3 clamp01( f3 c ) { return F3( clampf( c.x, 0.0f, 1.0f ), clampf( c.y, 0.0f, 1.0f ), clampf( c.z, 0.0f, 1.0f ) ); }
// colours are kept as f3 (r,g,b); every colour op clamps (CL/rt_funcs_math.cl:149-185)
DEVI f3 col_mult( f3 a, f3 b ) { return clamp01( F3( a.x * b.x, a.y * b.y, a.z * b.z ) ); }
DEVI f3 col_add( f3 a, f3 b ) { return clamp01( add3( a, b ) ); }
DEVI f3 col_scale( f3 a, float s ) { return clamp01( mul3( a, s ) ); }
DEVI f3 rgb_of( const rt_color &c ) { return F3( c.r, c.g, c.b ); }
// CL/rt_funcs_math.cl:119-124
DEVI f3 reflect3( f3 v, f3 n ) { float d = dot3( v, n ); return sub3( v, mul3( mul3( n, 2.0f ), d ) ); }
DEVI float minF( float a, float b ) { return ( a < b ) ? a : b; }
DEVI float maxF( float a, float b ) { return ( a > b ) ? a : b; }

DEVI uint32_t lane_id() { return threadIdx.x & 63u; }
DEVI uint32_t wave_bcast( uint32_t v ) { return (uint32_t) __builtin_amdgcn_readfirstlane( (int) v ); }
DEVI uint32_t lanes_below( unsigned long long mask )
{
	return __builtin_amdgcn_mbcnt_hi( (uint32_t) ( mask >> 32 ), __builtin_amdgcn_mbcnt_lo( (uint32_t) mask, 0u ) );
}

// Append through a wave-private reservation.  One atomic per 64-entry chunk
// on a list's end position is one atomic per chunk on ONE address, and the L2
// executes those one after the other (about 12 ns each): a pass over 130 000
// chunks cannot finish in under 1.5 ms whatever else it does.  A wave
// therefore reserves `block` entries at a time and hands them out locally.
// What is left of a reservation when the next batch does not fit, or when
// the wave exits, is filled with invalid entries (`mark`), which every
// consumer of the list skips; block = 0 reserves exactly what is needed (no
// holes, one atomic per batch), which is what short lists use.
#define RT_INVALID 0xffffffffu
struct Resv { uint32_t pos, end; };
template <class Mark>
DEVI void list_close( Resv &r, Mark mark )
{
	for ( uint32_t i = r.pos + lane_id(); i < r.end; i += 64u )
		mark( i );
	r.pos = r.end;
}
template <class Mark>
DEVI uint32_t list_append( uint32_t *counter, Resv &r, bool want, uint32_t block, Mark mark )
{
	const unsigned long long m = __ballot( want );
	const uint32_t n = (uint32_t) __popcll( m );
	if ( n == 0u )
		return 0u;
	if ( r.pos + n > r.end )
	{
		list_close( r, mark );
		const uint32_t take = block > n ? block : n;
		uint32_t base = 0;
		if ( lane_id() == 0u )
			base = atomicAdd( counter, take );
		base = wave_bcast( base );
		r.pos = base;
		r.end = base + take;
	}
	const uint32_t idx = r.pos + lanes_below( m );
	r.pos += n;
	return idx;
}
// reservation size for the output of a pass over `count` input entries
DEVI uint32_t resv_block( const Work &W, uint32_t count )
{
	if ( W.exactAppends )
		return 0u;
	return count >= gridDim.x * ( RT_BLOCK / 64u ) * 64u * 4u ? RT_RESV_BLOCK : 0u;
}

// LDS part of the traversal stacks of one workgroup: entry k of thread t sits
// at [k * RT_BLOCK + t], so a wave's push or pop touches 64 consecutive
// 8-byte words (no bank conflicts).  File scope keeps every access in the
// LDS address space.
__shared__ uint2 g_lds_stack[RT_LDS_STACK * RT_BLOCK];

// the part of a wave's traversal stacks beyond LDS: a strip of HBM, [entry][thread of the launch].  Only the
// wave-uniform base is kept (scalar registers); the lane's own address is worked out when an entry is spilled,
// which is rare (3 pushes in 10 000 on the benchmark frame)
struct StackRef { uint2 *base; uint32_t stride; };
DEVI uint2 *spill_entry( const StackRef &s, int sp )
{
	// (the thread number passes through an empty asm statement: otherwise base + thread is a loop-invariant
	// 64-bit address that occupies two vector registers for the whole launch)
	uint32_t t = threadIdx.x;
	asm volatile( "" : "+v"( t ) );
	return s.base + (size_t) ( sp - RT_LDS_STACK ) * s.stride + t;
}

// nodeSlots / triSlots: 64 lane-slots per wave-level iteration of the descend /
// leaf loop, shared among the lanes active in it (lane utilisation = nodes / nodeSlots)
struct Tally { uint32_t nodes, tris, spills, maxSp; float nodeSlots, triSlots; };
DEVI float lane_share() { return 64.0f / (float) __popcll( __ballot( 1 ) ); }

template <bool STATS>
DEVI void stack_push( const StackRef &s, int sp, uint32_t node, float tFar, Tally &tl )
{
	uint2 v = make_uint2( node, __float_as_uint( tFar ) );
	if ( sp < RT_LDS_STACK )
		g_lds_stack[sp * RT_BLOCK + threadIdx.x] = v;
	else
	{
		*spill_entry( s, sp ) = v;
		if ( STATS ) ++tl.spills;
	}
	if ( STATS && (uint32_t) ( sp + 1 ) > tl.maxSp ) tl.maxSp = sp + 1;
}
DEVI uint2 stack_pop( const StackRef &s, int sp )
{
	uint2 v;
	if ( sp < RT_LDS_STACK )
		v = g_lds_stack[sp * RT_BLOCK + threadIdx.x];
	else
	{
		// (volatile: or the compiler makes ONE load of the two, through a pointer that is either's - a flat load, slower
		// than an LDS read and waited for together with every record load in flight; entries out here are rare)
		const volatile uint32_t *q = (const volatile uint32_t *) spill_entry( s, sp );
		v.x = q[0];
		v.y = q[1];
	}
	return v;
}

// ---------------------------------------------------------------------------
// intersections (reference CL/rt_intersection.cl)
// ---------------------------------------------------------------------------

// :7-64
DEVI int ray_sphere( const Ray &r, float4 s, float &t )
{
	f3 tmp = sub3( r.src, F3( s.x, s.y, s.z ) );
	float b = dot3( r.dest, tmp );
	b *= 2.0f;
	float c = dot3( tmp, tmp );
	c -= s.w * s.w;
	float d = b * b - 4 * c;
	if ( d < 0.0f )
		return 0;
	d = sqrtf( d );
	float x1 = ( -b - d ) / 2.0f;
	float x2 = ( -b + d ) / 2.0f;
	if ( x2 < 0.0f )
		return 0;
	if ( x1 < 0.0f )
	{
		t = x2;
		return -1;
	}
	t = x1;
	return 1;
}

// Two-wide fp32 values: operations on them compile to the packed fp32
// instructions, which round each half exactly like the scalar instruction, so
// packing changes instruction counts, not results.
typedef float v2f __attribute__(( ext_vector_type( 2 ) ));
DEVI v2f V2( float x, float y ) { v2f r = { x, y }; return r; }
DEVI v2f splat( float x ) { v2f r = { x, x }; return r; }
DEVI v2f swap2( v2f a ) { return __builtin_shufflevector( a, a, 1, 0 ); }

// per-ray constants of the triangle test in the register pairs it consumes.
// Signs are folded into the constants (a * -b == -(a * b) and x + -y == x - y
// hold exactly), so that every difference of two products below is the plain
// sum of one register pair with another.
struct TriRay
{
	v2f dA;               // (-dest.x, dest.y)
	v2f dZ;               // ( dest.z, -dest.z)
	v2f sx, sy, sz;       // (src.c, -0): adding (-v0.c, e1.c) yields (src.c - v0.c, e1.c)
};
DEVI TriRay tri_ray( f3 src, f3 dest )
{
	TriRay r;
	r.dA = V2( -dest.x, dest.y );
	r.dZ = V2( dest.z, -dest.z );
	r.sx = V2( src.x, -0.0f );
	r.sy = V2( src.y, -0.0f );
	r.sz = V2( src.z, -0.0f );
	return r;
}
// keeps a value in a vector register of its own (stops the compiler from
// re-vectorising scalar results through extra moves)
DEVI float own_reg( float x ) { asm( "" : "+v"( x ) ); return x; }

// Correctly rounded 1 / x.  The hardware reciprocal is good to one unit in the
// last place; one Newton step carried out with two fused multiply-adds then
// lands on the correctly rounded quotient (Markstein) as long as neither x
// nor 1 / x leaves the normal range.  `exact` tells whether that held: the
// estimate is then a normal number.  Where it is false for any lane the
// caller takes the full IEEE division for the whole wave instead (zero
// determinants: degenerate or edge-on triangles).  rt_hip_selftest_rcp checks
// the claim for every one of the 2^32 float bit patterns on the device.
DEVI float rcp_newton( float x, bool &exact )
{
	const float r = __builtin_amdgcn_rcpf( x );
	exact = __builtin_amdgcn_classf( r, 0x108 );      // +/- normal
	const float e = __builtin_fmaf( -x, r, 1.0f );
	return __builtin_fmaf( r, e, r );
}
DEVI float rcp_exact( float x )
{
	bool ok;
	float r = rcp_newton( x, ok );
	if ( __ballot( !ok ) != 0ull )
	{
		// an empty volatile statement: without it the compiler evaluates the
		// division unconditionally and merely selects the result
		asm volatile( "" );
		r = 1.0f / x;
	}
	return r;
}

// one 48-byte leaf record as three 16-byte loads
struct Rec { float4 a, b, c; };
DEVI Rec load_rec( const float4 *recs, uint32_t i )
{
	const float4 *p = recs + 3u * (size_t) i;
	Rec r;
	r.a = p[0];
	r.b = p[1];
	r.c = p[2];
	return r;
}
DEVI uint32_t rec_tri( const Rec &r ) { return __float_as_uint( r.c.y ); }
DEVI uint32_t rec_mat( const Rec &r ) { return __float_as_uint( r.c.z ); }

// CL/rt_intersection.cl:110-154 with e1 / e2 taken from the leaf record.  Same
// values and the same accept / reject decisions as the reference's early-out
// form, evaluated without branches (a wave runs the long path anyway as soon
// as one lane needs it, and straight-line code lets the record loads be issued
// together) and two products at a time:
//   p = dest x e2, dist = src - v0, q = dist x e1,
//   det = e1.p, u = (p.dist)/det, v = (q.dest)/det, t = (q.e2)/det
// Every product and every sum below is one of the reference's, rounded once
// (x - y is also written x + (-y), which is the same operation).
DEVI bool ray_tri( const TriRay &R, const Rec &rc, float &t, float &u, float &v )
{
	const v2f e2xy = V2( rc.b.z, rc.b.w );
	const float e2z = rc.c.x;
	// dest x e2
	const v2f m1 = R.dA * splat( e2z );                  // (-dx*ez,  dy*ez)
	const v2f m2 = R.dZ * e2xy;                          // ( dz*ex, -dz*ey)
	const v2f m3 = R.dA * swap2( e2xy );                 // (-dx*ey,  dy*ex)
	const v2f n = m2 + m1;                               // (p.y, p.x)
	const float pz = -m3.x - m3.y;
	// (dist.c, e1.c) per component
	const v2f X = R.sx + V2( rc.a.x, rc.a.y );
	const v2f Y = R.sy + V2( rc.a.z, rc.a.w );
	const v2f Z = R.sz + V2( rc.b.x, rc.b.y );
	// (p.dist, p.e1) = ((px*X + py*Y) + pz*Z)
	const v2f ud = ( splat( n.y ) * X + splat( n.x ) * Y ) + splat( pz ) * Z;
	const float det = ud.y;
	const float inv = rcp_exact( det );
	// dist x e1
	const v2f c1 = Y * swap2( Z );                       // (ty*e1z, e1y*tz)
	const v2f c2 = Z * swap2( X );                       // (tz*e1x, e1z*tx)
	const v2f c3 = X * swap2( Y );                       // (tx*e1y, e1x*ty)
	const float qx = own_reg( c1.x - c1.y ), qy = own_reg( c2.x - c2.y ), qz = own_reg( c3.x - c3.y );
	const v2f qxy = V2( qx, qy );
	const v2f wd = qxy * R.dA, we = qxy * e2xy;          // (-qx*dx, qy*dy), (qx*ex, qy*ey)
	const float vn = own_reg( -wd.x + wd.y ) + qz * R.dZ.x;
	const float tn = own_reg( we.x + we.y ) + qz * e2z;
	u = ud.x * inv;
	v = vn * inv;
	t = tn * inv;
	bool reject = ( det == 0.0f ) | ( u < 0.0f ) | ( u > 1.0f ) | ( v < 0.0f ) | ( v + u > 1.0f );
	return !reject & ( t > 0.0f );
}

// :156-200
DEVI bool box_ray( const DevScene &S, const Ray &r, float &tNear, float &tFar )
{
	float t1 = ( S.bbLo[0] - r.src.x ) * r.inv.x;
	float t2 = ( S.bbHi[0] - r.src.x ) * r.inv.x;
	float t3 = ( S.bbLo[1] - r.src.y ) * r.inv.y;
	float t4 = ( S.bbHi[1] - r.src.y ) * r.inv.y;
	float t5 = ( S.bbLo[2] - r.src.z ) * r.inv.z;
	float t6 = ( S.bbHi[2] - r.src.z ) * r.inv.z;
	tNear = maxF( maxF( minF( t1, t2 ), minF( t3, t4 ) ), minF( t5, t6 ) );
	tFar = minF( minF( maxF( t1, t2 ), maxF( t3, t4 ) ), maxF( t5, t6 ) );
	if ( tFar < 0 )
		return false;
	if ( tNear > tFar )
		return false;
	return true;
}

// component `axis` of a vector held in registers.  Written on values, not on
// struct members behind a reference: selecting between member loads gets
// turned into a dynamically indexed load, which forces the ray into scratch.
DEVI float pick( uint32_t axis, float x, float y, float z ) { return axis == 0u ? x : ( axis == 1u ? y : z ); }

// inner-node step shared by both traversals (CL/rt_funcs_primitives.cl:185-251)
template <bool STATS>
DEVI uint32_t kd_inner( uint2 n, uint32_t cur, f3 src, f3 inv, f3 dest, float tNear, float &tFar,
	const StackRef &st, int &sp, Tally &tl )
{
	uint32_t axis = n.y & 3u;
	float sep = __uint_as_float( n.x );
	float o = pick( axis, src.x, src.y, src.z );
	float iv = pick( axis, inv.x, inv.y, inv.z );
	float d = pick( axis, dest.x, dest.y, dest.z );
	float tSplit = ( sep - o ) * iv;
	uint32_t nearN = cur + 1u, farN = n.y >> 2;
	if ( d < 0.0f )
	{
		uint32_t tmp = nearN;
		nearN = farN;
		farN = tmp;
	}
	if ( tSplit >= tFar )
		return nearN;
	if ( tSplit <= tNear )
		return farN;
	stack_push<STATS>( st, sp, farN, tFar, tl );
	++sp;
	tFar = tSplit;
	return nearN;
}

// Per-lane traversal state.  A traversal runs in rounds: descend to the next
// leaf, scan it, then either finish or pop.
struct Trav
{
	float tNear, tFar;
	uint32_t cur;
	int sp;
	uint2 n;          // node `cur`, already loaded
};

// A direction component of exactly zero (or so small that its reciprocal
// overflows) makes the split distance (sep - origin) * inv an infinity, or a
// NaN when the origin lies in the split plane.  A NaN compares false both
// ways, is taken as "both children" and becomes the near child's exit distance,
// where it rejects every hit; which cells get poisoned this way depends on the
// whole path, including the cells behind the origin.  Such rays (the middle
// row / column of an axis-aligned camera, mirror images of it) are therefore
// walked exactly as the reference walks them, without the shortcuts that are
// only equivalent for ordered distances.
DEVI bool axis_parallel( const Ray &r )
{
	return __builtin_isinf( r.inv.x ) || __builtin_isinf( r.inv.y ) || __builtin_isinf( r.inv.z );
}

// start of both queries (CL/rt_funcs_primitives.cl:160-163 / :289-291): scene
// box test; false = the ray misses the box
template <bool STATS>
DEVI bool trav_begin( const DevScene &S, const Ray &r, Trav &t, Tally &tl )
{
	if ( !box_ray( S, r, t.tNear, t.tFar ) )
		return false;
	// A ray that starts inside the scene box has tNear < 0; the reference walks
	// the cells behind the origin too (its box test does not clamp).  No hit
	// can be accepted there (t > 0 is required and a leaf only confirms hits
	// before its own exit distance), so start at the origin instead - unless the
	// ray is parallel to an axis (see axis_parallel).
	if ( t.tNear < 0.0f && !axis_parallel( r ) )
		t.tNear = 0.0f;
	t.cur = 0u;
	t.sp = 0;
	t.n = S.nodes[0];
	if ( STATS ) ++tl.nodes;
	return true;
}

// descend until `t.n` is a leaf; the whole wave stays in this tight loop until
// every active lane has reached one, so the leaf scans run together
template <bool STATS>
DEVI void trav_descend( const DevScene &S, f3 src, f3 inv, f3 dest, const StackRef &st, Trav &t, Tally &tl )
{
	while ( ( t.n.y & 3u ) != 3u )
	{
		t.cur = kd_inner<STATS>( t.n, t.cur, src, inv, dest, t.tNear, t.tFar, st, t.sp, tl );
		t.n = S.nodes[t.cur];
		if ( STATS ) { ++tl.nodes; tl.nodeSlots += lane_share(); }
	}
}

// leave the current leaf: false = stack empty (traversal over)
template <bool STATS>
DEVI bool trav_pop( const DevScene &S, const StackRef &st, Trav &t, Tally &tl )
{
	if ( !t.sp )
		return false;
	--t.sp;
	uint2 e = stack_pop( st, t.sp );
	t.cur = e.x;
	t.tNear = t.tFar;
	t.tFar = __uint_as_float( e.y );
	t.n = S.nodes[t.cur];
	if ( STATS ) ++tl.nodes;
	return true;
}

struct NearHit { float minT, minU, minV; uint32_t prN; int nearestB; };

// One triangle of the closest-hit leaf loop, CL/rt_funcs_primitives.cl:74-111.
template <bool STATS>
DEVI void nearest_tri( const TriRay &R, const Rec &rc, float &lt, NearHit &h, Tally &tl )
{
	float tt, u, v;
	if ( STATS ) { ++tl.tris; tl.triSlots += lane_share(); }
	if ( ray_tri( R, rc, tt, u, v ) && tt < lt )
	{
		lt = tt;
		h.minU = u;
		h.minV = v;
		h.prN = rec_tri( rc );
		h.nearestB = 1;
	}
}

// One triangle of the shadow query (sum of the opacities of every triangle hit
// before the light), CL/rt_funcs_primitives.cl:256-276.
template <bool STATS>
DEVI void alpha_tri( const DevScene &S, const TriRay &R, const Rec &rc, const Trav &t, float d, float &alpha, Tally &tl )
{
	float tt, u, v;
	if ( STATS ) { ++tl.tris; tl.triSlots += lane_share(); }
	if ( ray_tri( R, rc, tt, u, v ) && ( tt < d ) && ( tt < t.tFar ) && ( tt > t.tNear ) )
		alpha += S.mats[rec_mat( rc )].color.a;
}

// ---------------------------------------------------------------------------
// Cooperative walk: the rays a wave has left share its 64 lanes
//
// Rays are of very different length (a ray that skims the surface passes
// through dozens of leaves), so a wave that gives every ray one lane soon
// carries a few long rays in mostly idle lanes, one leaf per round - a chain of
// dependent loads that idle lanes do not shorten.  Once at most half the lanes
// are still walking, the wave changes gear: a ray owns a SEGMENT of q lanes
// (q = 64 / rays still walking, rounded down to a power of two, so it grows as
// rays finish) and advances up to q leaves per round:
//
//   seed     the segment's first lane takes the top of the ray's traversal
//            stack (far child + exit distance, exactly what the lane-per-ray
//            walk pushed; the entry distance is the exit distance of the last
//            leaf visited);
//   expand   every level, each lane replaces its inner node by the one or two
//            children the ray passes through, front to back, compacted within
//            the segment - the kd step of CL/rt_funcs_primitives.cl:185-251 applied to
//            all nodes of a level at once.  Entries that no longer fit go back
//            onto the ray's stack, farthest first, so the stack stays what the
//            reference's stack is: the not yet visited sub-trees, nearest on
//            top.  A segment whose entries are all leaves and that has room
//            takes more sub-trees off the stack.  At the end the segment holds
//            the ray's next leaves in the order the reference visits them, each
//            with the interval the reference's walk gives it;
//   scan     every lane tests the triangles of its leaf;
//   resolve  the reference's sequential rule over the leaves (first leaf whose
//            closest hit lies before the leaf's exit ends the ray; the hit flag
//            and the triangle of earlier leaves persist), evaluated with ballots
//            by the lane that owns the ray.
//
// Every node is still loaded once and every interval is the reference's, so
// hits are the same bit for bit; what changes is that a ray's leaves are
// scanned side by side and its sub-trees descended side by side.  Leaves
// behind the one that ends the ray are scanned in vain; that is the price.
//
// Only the LDS part of a stack is shared between lanes.  A ray whose stack
// does not fit there is RESTARTED instead: the walk begins again at the root
// over [T, end], T = exit distance of the last leaf scanned.  With
// V(node, a, b) = the leaf sequence of the reference's walk of `node` over
// [a, b] (near only if split >= b, far only if split <= a, else V(near, a,
// split) then V(far, split, b)), every boundary T between two consecutive
// leaves is the split distance of an ancestor at which both children were
// taken, and V(root, T, b) is the rest of V(root, a, b) behind that boundary
// with identical intervals: at nodes where T lies in the near part the
// comparisons come out as before (a < T < split), at the node whose split IS
// T the walk takes the far child only (split <= T), at nodes whose far part
// contains T likewise.  This needs ordered distances, so rays parallel to an
// axis (see axis_parallel) never walk cooperatively.
// ---------------------------------------------------------------------------

#ifdef RT_PROFILE_CHUNKS
struct Prof { uint32_t lock, coop, iters; unsigned long long tExpand, tScan, tCoop, tDescend, tLeaf; };
#define PROF_T0( v ) const unsigned long long v = wall_clock64()
#define PROF_ADD( f, v ) ( pf.f += wall_clock64() - v )
#define PROF_ARG , Prof &pf
#define PROF_PASS , pf
#define PROF_INC( x ) ( ++pf.x )
#else
#define PROF_ARG
#define PROF_PASS
#define PROF_INC( x ) ( (void) 0 )
#define PROF_T0( v ) ( (void) 0 )
#define PROF_ADD( f, v ) ( (void) 0 )
#endif
// exchange area of the expand step: one (node, exit distance) pair per lane
__shared__ uint2 g_lds_xch[RT_BLOCK];
// rank -> lane of the rays still walking, one table per wave
__shared__ uint32_t g_lds_owner[RT_BLOCK];

DEVI float shfl_f( float x, uint32_t lane ) { return __uint_as_float( (uint32_t) __builtin_amdgcn_ds_bpermute( (int) ( lane << 2 ), (int) __float_as_uint( x ) ) ); }
DEVI uint32_t shfl_u( uint32_t x, uint32_t lane ) { return (uint32_t) __builtin_amdgcn_ds_bpermute( (int) ( lane << 2 ), (int) x ); }
// LDS traffic between the lanes of one wave: program order is enough for the
// hardware (a wave's LDS instructions execute in order); this keeps the
// compiler from moving the accesses across
DEVI void wave_lds_sync()
{
	__builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
	__builtin_amdgcn_wave_barrier();
}

// a lane's place in its segment
struct Seg
{
	uint32_t q, start, j;                 // lanes per segment, first lane of mine, my index in it
	unsigned long long mask;              // lanes of my segment
};
DEVI Seg make_seg( uint32_t q )
{
	Seg g;
	const uint32_t l = lane_id();
	g.q = q;
	g.start = l & ~( q - 1u );
	g.j = l - g.start;
	g.mask = q >= 64u ? ~0ull : ( ( ( 1ull << q ) - 1ull ) << g.start );
	return g;
}

// What a lane holds after the expand step: a leaf (its two node words) and the
// interval of the ray inside it.
struct SegLeaf { uint2 n; float tN, tF; };

// the value the lane before mine holds (lane 0: unspecified)
DEVI float lane_before( float x )
{
	return __uint_as_float( (uint32_t) __builtin_amdgcn_update_dpp( 0, (int) __float_as_uint( x ), 0x138 /* wave_shr:1 */, 0xf, 0xf, false ) );
}

// Expand (see above).  In: the ray of my segment (sr), its entry distance T,
// the number of entries on its stack (sp, in the LDS column of thread
// `col`), restart = begin at the root over [T, tEnd] instead.  Out: lanes with
// `have` hold a leaf each, in walk order; sp = what is left on the stack;
// lost = the stack did not fit (its content is void, the ray restarts).
//
// A segment and the stack behind it are ONE sequence of sub-trees in walk order
// whose intervals follow each other without gaps (the exit distance of one is
// the entry distance of the next; the first one starts at T).  An entry is
// therefore a (node, exit distance) pair, the format of a stack entry, and an
// iteration is: every lane replaces its node by the children the ray passes
// through (a leaf stays), the new sequence is written in order - the first q
// entries to the exchange area, the rest back onto the stack, farthest first -,
// and every lane reads the entry of its position; positions behind the end of
// the sequence are filled from the top of the stack.  The node words are
// loaded after the exchange.
template <bool STATS>
DEVI void seg_expand( const DevScene &S, const Seg &g, const Ray &sr, bool segOn, float T, float tEnd, bool restart, uint32_t col, uint32_t cap,
	SegLeaf &lf, bool &have, uint32_t &sp, bool &lost, Tally &tl PROF_ARG )
{
	uint2 *const xs = g_lds_xch + ( threadIdx.x - g.j );           // slot 0 of my segment
	uint2 *const stk = g_lds_stack + col;                          // entry k of the ray's stack: stk[k * RT_BLOCK]
	lost = false;
	have = segOn && restart && g.j == 0u;                          // a restarted ray begins with the root
	if ( restart || !segOn )
		sp = 0u;
	uint32_t cur = 0u, kept = segOn && restart ? 1u : 0u;
	float tN = T, tF = tEnd;
	uint2 n = make_uint2( 0u, 3u );
	if ( have )
	{
		n = S.nodes[0];
		if ( STATS ) ++tl.nodes;
	}
	do
	{
		PROF_INC( iters );
		// the kd step of my node, CL/rt_funcs_primitives.cl:185-251
		const bool inner = have && ( n.y & 3u ) != 3u;
		const uint32_t axis = n.y & 3u;
		const float sep = __uint_as_float( n.x );
		const float o = pick( axis, sr.src.x, sr.src.y, sr.src.z );
		const float iv = pick( axis, sr.inv.x, sr.inv.y, sr.inv.z );
		const float d = pick( axis, sr.dest.x, sr.dest.y, sr.dest.z );
		const float tSplit = ( sep - o ) * iv;
		const bool back = d < 0.0f;
		const uint32_t left = cur + 1u, right = n.y >> 2;
		const uint32_t nearN = back ? right : left, farN = back ? left : right;
		const bool onlyNear = tSplit >= tF;
		const bool onlyFar = !onlyNear && tSplit <= tN;
		const bool two = inner && !onlyNear && !onlyFar;
		// bit 31 marks an entry whose node has not been counted yet (counting build)
		const uint32_t fresh = STATS ? 0x80000000u : 0u;
		const uint32_t c0 = inner ? ( ( onlyFar ? farN : nearN ) | fresh ) : cur;
		const float b0 = two ? tSplit : tF;
		// my place in the new sequence
		// (the lanes that hold an entry are the first `kept` of the segment)
		const unsigned long long m2 = __ballot( two );
		const uint32_t pos = g.j + lanes_below( m2 & g.mask );         // (those of my segment before me)
		const uint32_t total = kept + (uint32_t) __popcll( m2 & g.mask );
		const uint32_t inSeg = total < g.q ? total : g.q;
		const uint32_t over = total - inSeg;
		uint32_t take = g.q - inSeg;                                  // free lanes take sub-trees off the stack
		if ( take > sp ) take = sp;
		if ( sp + over > cap )
			lost = true;
		const uint32_t top = sp + total - 1u;                         // position p >= q goes to stack entry top - p
		if ( have && ( pos < g.q || !lost ) )
		{
			uint2 *const w = pos < g.q ? xs + pos : stk + ( top - pos ) * RT_BLOCK;
			*w = make_uint2( c0, __float_as_uint( b0 ) );
		}
		if ( two && ( pos + 1u < g.q || !lost ) )
		{
			uint2 *const w = pos + 1u < g.q ? xs + pos + 1u : stk + ( top - pos - 1u ) * RT_BLOCK;
			*w = make_uint2( farN | fresh, __float_as_uint( tF ) );
		}
		wave_lds_sync();
		kept = inSeg + take;
		have = g.j < kept;
		uint2 e = make_uint2( 0u, 0u );
		if ( have )
			e = g.j < inSeg ? xs[g.j] : stk[( sp - 1u - ( g.j - inSeg ) ) * RT_BLOCK];
		wave_lds_sync();
		if ( STATS && have && ( g.j >= inSeg || ( e.x >> 31 ) ) ) ++tl.nodes;
		cur = STATS ? e.x & 0x7fffffffu : e.x;
		tF = __uint_as_float( e.y );
		sp = lost ? 0u : sp + over - take;
		tN = lane_before( tF );
		if ( g.j == 0u )
			tN = T;
		n = make_uint2( 0u, 3u );
		if ( have )
			n = S.nodes[cur];
	}
	while ( __ballot( have && ( n.y & 3u ) != 3u ) != 0ull );
	lf.n = n;
	lf.tN = tN;
	lf.tF = tF;
}

// Continues the closest-hit query of every lane with `active` set
// cooperatively (see above).  In: the lane is about to open node t.cur over
// [t.tNear, t.tFar] with t.sp entries on its stack; h carries the persistent
// part of the hit state in and the result out.  Wave-uniform.
template <bool STATS>
DEVI void nearest_coop( const DevScene &S, const Ray &r, bool active, const Trav &t, NearHit &h, Tally &tl, uint32_t cap PROF_ARG )
{
	uint32_t *const own = g_lds_owner + ( threadIdx.x & ~63u );
	// from here on a ray is (entry distance T, stack): the node the lane was about
	// to open goes onto its stack
	uint32_t sp = 0u;
	bool restart = false;
	float T = t.tNear;
	if ( active )
	{
		if ( (uint32_t) t.sp < cap )
		{
			g_lds_stack[t.sp * RT_BLOCK + threadIdx.x] = make_uint2( t.cur, __float_as_uint( t.tFar ) );
			sp = (uint32_t) t.sp + 1u;
		}
		else
			restart = true;                  // part of this stack lives outside LDS
	}
	wave_lds_sync();
	for ( ;; )
	{
		const unsigned long long mA = __ballot( active );
		if ( mA == 0ull )
			break;
		const uint32_t nA = (uint32_t) __popcll( mA );
		PROF_INC( coop );
		const uint32_t lq = 31u - (uint32_t) __clz( (int) ( 64u / nA ) ), q = 1u << lq;
		const Seg g = make_seg( q );
		// segment s belongs to the s-th walking ray
		const uint32_t rank = lanes_below( mA );
		if ( active )
			own[rank] = lane_id();
		wave_lds_sync();
		const uint32_t seg = lane_id() >> lq;
		const bool segOn = seg < nA;
		const uint32_t owner = segOn ? own[seg] : lane_id();
		wave_lds_sync();
		Ray sr;
		sr.src = F3( shfl_f( r.src.x, owner ), shfl_f( r.src.y, owner ), shfl_f( r.src.z, owner ) );
		sr.dest = F3( shfl_f( r.dest.x, owner ), shfl_f( r.dest.y, owner ), shfl_f( r.dest.z, owner ) );
		sr.inv = F3( shfl_f( r.inv.x, owner ), shfl_f( r.inv.y, owner ), shfl_f( r.inv.z, owner ) );
		const float sT = shfl_f( T, owner );
		uint32_t sSp = shfl_u( sp, owner );
		const bool sRestart = shfl_u( restart ? 1u : 0u, owner ) != 0u;
		float sEnd = 0.0f;
		if ( __ballot( sRestart ) != 0ull )
		{
			float tn;
			(void) box_ray( S, sr, tn, sEnd );                 // the ray did hit the box when it started
		}

		SegLeaf en;
		bool have, lost;
		PROF_T0( pe );
		seg_expand<STATS>( S, g, sr, segOn, sT, sEnd, sRestart, ( threadIdx.x & ~63u ) + owner, cap, en, have, sSp, lost, tl PROF_PASS );
		PROF_ADD( tExpand, pe );
		PROF_T0( ps );

		// scan: my leaf, CL/rt_funcs_primitives.cl:74-111
		const TriRay R = tri_ray( sr.src, sr.dest );
		float lt = INFINITY;
		NearHit lh = { INFINITY, 0.0f, 0.0f, 0u, 0 };
		const uint32_t cnt = have ? en.n.y >> 2 : 0u;
		if ( cnt != 0u )
		{
			uint32_t i = en.n.x;
			const uint32_t end = i + cnt;
			Rec r0 = load_rec( S.recs, i );
			for ( ;; )
			{
				Rec r1 = load_rec( S.recs, i + 1u );
				nearest_tri<STATS>( R, r0, lt, lh, tl );
				if ( ++i == end ) break;
				r0 = load_rec( S.recs, i + 1u );
				nearest_tri<STATS>( R, r1, lt, lh, tl );
				if ( ++i == end ) break;
			}
		}

		PROF_ADD( tScan, ps );
		// resolve, by the lane that owns the ray: the leaves of its segment in
		// order (:168-183).  The first leaf with a hit before its exit distance
		// ends the ray; otherwise the last leaf with any hit leaves its triangle
		// and barycentrics behind, the running distance is that of the last leaf
		// scanned, and the ray goes on with what is on its stack.
		const unsigned long long bTerm = __ballot( have && lh.nearestB && lt < en.tF );
		const unsigned long long bHit = __ballot( have && lh.nearestB != 0 );
		const unsigned long long bHave = __ballot( have ), bLost = __ballot( lost );
		const uint32_t s0 = rank << lq;                                  // first lane of my ray's segment
		const unsigned long long sm = q >= 64u ? ~0ull : ( ( 1ull << q ) - 1ull );
		const unsigned long long mTerm = ( bTerm >> s0 ) & sm, mHit = ( bHit >> s0 ) & sm;
		const uint32_t nKept = (uint32_t) __popcll( ( bHave >> s0 ) & sm );
		const uint32_t last = s0 + ( nKept ? nKept - 1u : 0u );
		const uint32_t srcT = s0 + ( mTerm ? (uint32_t) __ffsll( (long long) mTerm ) - 1u : 0u );
		const uint32_t srcH = s0 + ( mHit ? 63u - (uint32_t) __clzll( (long long) mHit ) : 0u );
		const uint32_t src = ( mTerm ? srcT : srcH ) & 63u;
		const float vU = shfl_f( lh.minU, src ), vV = shfl_f( lh.minV, src );
		const uint32_t vN = shfl_u( lh.prN, src );
		const float vT = shfl_f( lt, ( mTerm ? srcT : last ) & 63u );
		const float lastF = shfl_f( en.tF, last & 63u );
		const uint32_t spLeft = shfl_u( sSp, s0 & 63u );
		if ( active )
		{
			if ( mTerm | mHit )
			{
				h.minU = vU;
				h.minV = vV;
				h.prN = vN;
				h.nearestB = 1;
			}
			h.minT = vT;
			restart = ( ( bLost >> s0 ) & 1ull ) != 0ull;
			sp = spLeft;
			T = lastF;
			if ( mTerm || ( sp == 0u && !restart ) )
				active = false;
		}
	}
}

// The shadow query walked cooperatively (see "Cooperative walk"): same expand
// step; every lane collects the hits of its leaf that count
// (CL/rt_funcs_primitives.cl:256-276), and the opacities are then added to the
// ray's sum in the reference's order - leaf after leaf, triangle after
// triangle - by passing the running sum through the lanes of the segment that
// have hits.  In: the lane is about to open node t.cur over [t.tNear, t.tFar]
// with t.sp entries on its stack, alpha = the sum so far.
template <bool STATS>
DEVI void alpha_coop( const DevScene &S, const Ray &r, bool active, const Trav &t, float d, float &alpha, Tally &tl, uint32_t cap PROF_ARG )
{
	uint32_t *const own = g_lds_owner + ( threadIdx.x & ~63u );
	uint32_t sp = 0u;
	bool restart = false;
	float T = t.tNear;
	// the walk ends at the light; a restarted walk must end there too
	const float tStop = t.tFar;
	(void) tStop;
	if ( active )
	{
		if ( (uint32_t) t.sp < cap )
		{
			g_lds_stack[t.sp * RT_BLOCK + threadIdx.x] = make_uint2( t.cur, __float_as_uint( t.tFar ) );
			sp = (uint32_t) t.sp + 1u;
		}
		else
			restart = true;
	}
	wave_lds_sync();
	for ( ;; )
	{
		const unsigned long long mA = __ballot( active );
		if ( mA == 0ull )
			break;
		const uint32_t nA = (uint32_t) __popcll( mA );
		PROF_INC( coop );
		const uint32_t lq = 31u - (uint32_t) __clz( (int) ( 64u / nA ) ), q = 1u << lq;
		const Seg g = make_seg( q );
		const uint32_t rank = lanes_below( mA );
		if ( active )
			own[rank] = lane_id();
		wave_lds_sync();
		const uint32_t seg = lane_id() >> lq;
		const bool segOn = seg < nA;
		const uint32_t owner = segOn ? own[seg] : lane_id();
		wave_lds_sync();
		Ray sr;
		sr.src = F3( shfl_f( r.src.x, owner ), shfl_f( r.src.y, owner ), shfl_f( r.src.z, owner ) );
		sr.dest = F3( shfl_f( r.dest.x, owner ), shfl_f( r.dest.y, owner ), shfl_f( r.dest.z, owner ) );
		sr.inv = F3( shfl_f( r.inv.x, owner ), shfl_f( r.inv.y, owner ), shfl_f( r.inv.z, owner ) );
		const float sT = shfl_f( T, owner ), sD = shfl_f( d, owner );
		float run = shfl_f( alpha, owner );
		uint32_t sSp = shfl_u( sp, owner );
		const bool sRestart = shfl_u( restart ? 1u : 0u, owner ) != 0u;
		float sEnd = 0.0f;
		if ( __ballot( sRestart ) != 0ull )
		{
			// as at the start of the query: the scene box, cut at the light
			float tn;
			(void) box_ray( S, sr, tn, sEnd );
			if ( sD < sEnd ) sEnd = sD;
		}

		SegLeaf en;
		bool have, lost;
		seg_expand<STATS>( S, g, sr, segOn, sT, sEnd, sRestart, ( threadIdx.x & ~63u ) + owner, cap, en, have, sSp, lost, tl PROF_PASS );

		// scan: the hits of my leaf that count, in triangle order; the first two
		// opacities are kept, a leaf with more is scanned again when its turn comes
		const TriRay R = tri_ray( sr.src, sr.dest );
		uint32_t nh = 0u;
		float a0 = 0.0f, a1 = 0.0f;
		const uint32_t cnt = have ? en.n.y >> 2 : 0u;
		for ( uint32_t i = 0u; i < cnt; ++i )
		{
			const Rec rc = load_rec( S.recs, en.n.x + i );
			float tt, u, v;
			if ( STATS ) { ++tl.tris; tl.triSlots += lane_share(); }
			if ( ray_tri( R, rc, tt, u, v ) && ( tt < sD ) && ( tt < en.tF ) && ( tt > en.tN ) )
			{
				const float a = S.mats[rec_mat( rc )].color.a;
				if ( nh == 0u ) a0 = a;
				else if ( nh == 1u ) a1 = a;
				++nh;
			}
		}

		// resolve: the running sum goes through the lanes with hits, in order
		unsigned long long mine = __ballot( nh != 0u ) & g.mask;
		while ( __ballot( mine != 0ull ) != 0ull )
		{
			const uint32_t turn = mine ? (uint32_t) __ffsll( (long long) mine ) - 1u : lane_id();
			float sum = run;
			if ( mine != 0ull && lane_id() == turn )
			{
				if ( nh <= 2u )
				{
					sum += a0;
					if ( nh == 2u ) sum += a1;
				}
				else
				{
					for ( uint32_t i = 0u; i < cnt; ++i )
					{
						const Rec rc = load_rec( S.recs, en.n.x + i );
						float tt, u, v;
						if ( ray_tri( R, rc, tt, u, v ) && ( tt < sD ) && ( tt < en.tF ) && ( tt > en.tN ) )
							sum += S.mats[rec_mat( rc )].color.a;
					}
				}
			}
			run = shfl_f( sum, turn );
			mine &= mine - 1ull;
		}

		const unsigned long long bHave = __ballot( have ), bLost = __ballot( lost );
		const uint32_t s0 = rank << lq;
		const unsigned long long sm = q >= 64u ? ~0ull : ( ( 1ull << q ) - 1ull );
		const uint32_t nKept = (uint32_t) __popcll( ( bHave >> s0 ) & sm );
		const uint32_t last = s0 + ( nKept ? nKept - 1u : 0u );
		const float lastF = shfl_f( en.tF, last & 63u );
		const uint32_t spLeft = shfl_u( sSp, s0 & 63u );
		const float total = shfl_f( run, s0 & 63u );
		if ( active )
		{
			alpha = total;
			restart = ( ( bLost >> s0 ) & 1ull ) != 0ull;
			sp = spLeft;
			T = lastF;
			// the caller clamps the sum to [0,1]: with non-negative opacities it cannot
			// drop below 1 again
			if ( ( S.alphaMonotone && alpha >= 1.0f ) || ( sp == 0u && !restart ) )
				active = false;
		}
	}
}

// Closest-hit query of every lane with `live` set, CL/rt_funcs_primitives.cl:165-252:
// rounds of descend / leaf scan / pop.  The per-leaf reset of the running
// distance while the hit flag and the triangle id persist is the reference's
// behaviour.  Control flow is wave-uniform, so that lanes whose ray is
// finished stay available for the cooperative scans.
//
// coopLive: once at most that many lanes are still walking (and at least one
// round has been walked, or `atOnce`), the rest of the rays is walked
// cooperatively (nearest_coop); 0 = never.  Rays parallel to an axis always
// keep their own lane, and the gear change waits for them.
template <bool STATS>
DEVI void nearest_wave( const DevScene &S, const Ray &r, const TriRay &R, const StackRef &st, bool live, Trav &t, NearHit &h, Tally &tl,
	uint32_t coopLive, uint32_t coopCap, bool atOnce PROF_ARG )
{
	const f3 src = r.src, inv = r.inv, dest = r.dest;
	// the cooperative walk is the last thing a chunk does (nothing of the lane-per-ray
	// walk stays live across it): rays parallel to an axis are finished first
	const bool axisPar = axis_parallel( r );
	const bool ownLane = coopLive == 0u || axisPar;
	// a ray parallel to an axis scans a leaf's COMPLETE list where the tree carries pruned ones: under the
	// reference's NaN split distances (see axis_parallel) a hit can be confirmed by a leaf that lists the
	// triangle without containing it
	const bool viaFull = axisPar && S.leafFull != nullptr;
	bool walked = atOnce;
	while ( __ballot( live ) != 0ull )
	{
		if ( walked && (uint32_t) __popcll( __ballot( live ) ) <= coopLive && __ballot( live && ownLane ) == 0ull )
		{
			PROF_T0( pc );
			nearest_coop<STATS>( S, r, live, t, h, tl, coopCap PROF_PASS );
			PROF_ADD( tCoop, pc );
			return;
		}
		walked = true;
		PROF_INC( lock );
		float lt = INFINITY;
		uint32_t cnt = 0u, first = 0u;
		PROF_T0( pd );
		if ( live )
		{
			trav_descend<STATS>( S, src, inv, dest, st, t, tl );
			first = t.n.x;
			cnt = t.n.y >> 2;
			if ( viaFull )
			{
				const uint2 f = S.leafFull[t.cur];
				first = f.x;
				cnt = f.y;
			}
		}
		PROF_ADD( tDescend, pd );
		PROF_T0( pl );
		if ( live && cnt != 0u )
		{
			// Two records are kept in flight and take turns (no register copies); the
			// array ends with one spare record, so the look-ahead load of the last
			// triangle stays inside the buffer.
			uint32_t i = first;
			const uint32_t end = i + cnt;
			Rec r0 = load_rec( S.recs, i );
			for ( ;; )
			{
				Rec r1 = load_rec( S.recs, i + 1u );
				nearest_tri<STATS>( R, r0, lt, h, tl );
				if ( ++i == end ) break;
				r0 = load_rec( S.recs, i + 1u );
				nearest_tri<STATS>( R, r1, lt, h, tl );
				if ( ++i == end ) break;
			}
		}
		PROF_ADD( tLeaf, pl );
		if ( live )
		{
			h.minT = lt;
			if ( h.nearestB && ( h.minT < t.tFar ) )
				live = false;
			else
				live = trav_pop<STATS>( S, st, t, tl );
		}
	}
}

// shadow query of every lane with `live` set, CL/rt_funcs_primitives.cl:295-378
template <bool STATS>
DEVI void alpha_wave( const DevScene &S, const Ray &r, const TriRay &R, const StackRef &st, bool live, Trav &t, float d, float &alpha, Tally &tl,
	uint32_t coopLive, uint32_t coopCap, bool atOnce PROF_ARG )
{
	const f3 src = r.src, inv = r.inv, dest = r.dest;
	const bool axisPar = axis_parallel( r );
	const bool ownLane = coopLive == 0u || axisPar;
	const bool viaFull = axisPar && S.leafFull != nullptr;      // (as in nearest_wave)
	bool walked = atOnce;
	while ( __ballot( live ) != 0ull )
	{
		if ( walked && (uint32_t) __popcll( __ballot( live ) ) <= coopLive && __ballot( live && ownLane ) == 0ull )
		{
			alpha_coop<STATS>( S, r, live, t, d, alpha, tl, coopCap PROF_PASS );
			return;
		}
		walked = true;
		PROF_INC( lock );
		uint32_t cnt = 0u, first = 0u;
		if ( live )
		{
			trav_descend<STATS>( S, src, inv, dest, st, t, tl );
			first = t.n.x;
			cnt = t.n.y >> 2;
			if ( viaFull )
			{
				const uint2 f = S.leafFull[t.cur];
				first = f.x;
				cnt = f.y;
			}
		}
		if ( live && cnt != 0u )
		{
			uint32_t i = first;
			const uint32_t end = i + cnt;
			Rec r0 = load_rec( S.recs, i );
			for ( ;; )
			{
				Rec r1 = load_rec( S.recs, i + 1u );
				alpha_tri<STATS>( S, R, r0, t, d, alpha, tl );
				if ( ++i == end ) break;
				r0 = load_rec( S.recs, i + 1u );
				alpha_tri<STATS>( S, R, r1, t, d, alpha, tl );
				if ( ++i == end ) break;
			}
		}
		if ( live )
		{
			// the caller clamps the sum to [0,1]: with non-negative opacities it
			// cannot drop below 1 again, the rest of the ray is irrelevant
			if ( S.alphaMonotone && alpha >= 1.0f )
				live = false;
			else
				live = trav_pop<STATS>( S, st, t, tl );
		}
	}
}

// ---------------------------------------------------------------------------
// rays
// ---------------------------------------------------------------------------

// camera ray of pixel (x, y) of the full image, CL/rt_raytrace.cl:27-28,60-79
DEVI Ray primary_ray( const DevScene &S, int x, int y )
{
	const float *W = S.cam, *P = S.cam + 16;
	Ray r;
	f3 d = F3( S.xTr * (float) x - 1.0f, S.yTr * (float) y - 1.0f, 1.0f );
	r.src = F3( S.camSrc[0], S.camSrc[1], S.camSrc[2] );
	d = F3( P[0] * d.x + P[4] * d.y + P[8] * d.z, P[1] * d.x + P[5] * d.y + P[9] * d.z,
	        P[2] * d.x + P[6] * d.y + P[10] * d.z );
	d = F3( W[0] * d.x + W[4] * d.y + W[8] * d.z, W[1] * d.x + W[5] * d.y + W[9] * d.z,
	        W[2] * d.x + W[6] * d.y + W[10] * d.z );
	r.dest = norm3( d );
	r.inv = rcp3( r.dest );
	return r;
}

// entry e of slot 0 = lane (e & 63) of the 8x8 pixel tile (e >> 6)
// ONE_TILE: the calling lanes hold consecutive entries of one chunk, i.e. of one tile (the closest-hit pass)
template <bool ONE_TILE>
DEVI bool slot0_pixel( const DevScene &S, const Work &W, uint32_t e, int &x, int &yLocal, int &yGlobal )
{
	// (the tile number of a chunk is wave-uniform and is divided in scalar registers; the band height passes
	// through an empty asm statement so that the constants of the division by it are worked out here, per chunk,
	// instead of being kept in vector registers for the whole launch)
	const uint32_t tile = ONE_TILE ? wave_bcast( e >> 6 ) : e >> 6, ln = e & 63u;
	const uint32_t ty = tile / W.tilesX, tx = tile - ty * W.tilesX;
	x = (int) ( tx * 8u + ( ln & 7u ) );
	yLocal = (int) ( ty * 8u + ( ln >> 3 ) );
	int bandH = S.bandH;
	asm volatile( "" : "+s"( bandH ) );
	int band = yLocal / bandH;
	yGlobal = ( band * S.nParts + S.part ) * bandH + ( yLocal - band * bandH );
	return x < S.w && yLocal < S.localH && yGlobal < S.h;
}

// Head of a ray-list entry: what a traversal needs.  pk = parent slot | type << 8
// (type 1 = reflection, 0 = refraction); slot 0 has no parent (0xff).
struct RayHead
{
	Ray ray;
	uint32_t pix, pk;
};

DEVI int parent_of( uint32_t pk ) { return ( pk & 0xffu ) == 0xffu ? -1 : (int) ( pk & 0xffu ); }

template <bool ONE_TILE = false>
DEVI bool load_ray( const DevScene &S, const Work &W, int slot, uint32_t e, uint32_t count, RayHead &rh )
{
	if ( slot == 0 )
	{
		int x, yl, yg;
		if ( !slot0_pixel<ONE_TILE>( S, W, e, x, yl, yg ) )
			return false;
		rh.ray = primary_ray( S, x, yg );
		rh.pix = (uint32_t) yl * (uint32_t) S.w + (uint32_t) x;
		rh.pk = 0x1ffu;
		return true;
	}
	if ( e >= count )
		return false;
	size_t o = (size_t) ( slot - 1 ) * W.cap + e;
	float4 a = W.rayA[o];
	if ( __float_as_uint( a.w ) == RT_INVALID )       // hole left by a reservation
		return false;
	float4 b = W.rayB[o];
	rh.ray.src = F3( a.x, a.y, a.z );
	rh.ray.dest = F3( b.x, b.y, b.z );
	rh.ray.inv = rcp3( rh.ray.dest );
	rh.pix = __float_as_uint( a.w );
	rh.pk = __float_as_uint( b.w );
	return true;
}

// weight with which this slot's colour is folded into its parent
// (CL/rt_funcs_primitives.cl:693-720); P is this slot's hit point.
// rayC = reflection: parent reflect rgb / refraction: parent color rgb * lightFalloff,
// rayD = parent hit point (the refraction weight needs the distance).
DEVI f3 fold_weight( const Work &W, int slot, uint32_t e, uint32_t pk, f3 P )
{
	size_t o = (size_t) ( slot - 1 ) * W.cap + e;
	float4 c = W.rayC[o];
	if ( ( pk >> 8 ) & 1u )
		return F3( c.x, c.y, c.z );
	float4 d = W.rayD[o];
	float dist = len3( sub3( F3( d.x, d.y, d.z ), P ) );
	return F3( cl_exp( c.x * -dist ), cl_exp( c.y * -dist ), cl_exp( c.z * -dist ) );
}

DEVI void flush_tally( const Work &W, const Tally &tl, bool shadow )
{
	atomicAdd( &W.ctr->nodeVisits, (unsigned long long) tl.nodes );
	atomicAdd( &W.ctr->triTests, (unsigned long long) tl.tris );
	atomicAdd( shadow ? &W.ctr->nodeSlotsShadow : &W.ctr->nodeSlots, (unsigned long long) tl.nodeSlots );
	atomicAdd( shadow ? &W.ctr->triSlotsShadow : &W.ctr->triSlots, (unsigned long long) tl.triSlots );
	if ( shadow )
	{
		atomicAdd( &W.ctr->nodeVisitsShadow, (unsigned long long) tl.nodes );
		atomicAdd( &W.ctr->triTestsShadow, (unsigned long long) tl.tris );
	}
	atomicAdd( &W.ctr->spillPushes, (unsigned long long) tl.spills );
	atomicMax( &W.ctr->maxStack, (unsigned long long) tl.maxSp );
}

DEVI StackRef make_stack( const Work &W )
{
	StackRef st;
	st.base = W.spill + (size_t) blockIdx.x * RT_BLOCK;
	st.stride = W.spillStride;
	return st;
}

// Work distribution over the resident waves of a launch, in 64-entry chunks.
// Atomics on ONE address are executed strictly one after the other by the L2
// (about 12 ns each on this chip): one atomic per chunk bounds a pass from
// below by chunks x 12 ns, and a launch in which every wave first has to ask
// for work starts with waves x 12 ns (75 us) of queueing.  So:
//   - a first share of the list is dealt out without atomics, round by round:
//     chunk = round * waves + wave (neighbouring waves, neighbouring chunks);
//   - the rest (`dynShare` eighths) goes through the counter and evens out
//     the waves.  The streaming passes, whose chunks all cost the same, keep
//     that rest small.  The traversal passes deal out only the first round
//     statically: ray cost follows the image (background against silhouette)
//     and a fixed wave -> chunk pattern follows it too (measured: 6/8 static
//     makes the frame 40 % slower);
//   - lists of at most two chunks per wave are dealt out entirely statically.
// Measured alternatives that lost: looking at the counter with a plain load
// before asking (loads of a line under atomic fire take far longer than the
// atomic: frame 2.6x slower); several consecutive chunks per atomic in the
// traversal passes (2 chunks +3 %, 8 chunks +27 %); narrower chunks for short
// lists (the long rays of the deep slots are not shortened by them).
struct Grab { uint32_t round; };
DEVI bool next_chunk( uint32_t *head, unsigned long long total, Grab &g, uint32_t &first, uint32_t dynShare, uint32_t shift = 6u )
{
	// a chunk is 2^shift entries (64, one per lane, except in the cooperative pass)
	const uint32_t chunks = (uint32_t) ( ( total + ( 1ull << shift ) - 1ull ) >> shift );
	const uint32_t waves = gridDim.x * ( RT_BLOCK / 64u );
	// (wave-uniform, and said so: a value derived from threadIdx counts as per-lane and would take a vector register
	// for the whole launch)
	const uint32_t wave = wave_bcast( blockIdx.x * ( RT_BLOCK / 64u ) + ( threadIdx.x >> 6 ) );
	// dynShare = 8: one static round, the rest through the counter
	const uint32_t rounds = chunks <= 2u * waves ? ( chunks + waves - 1u ) / waves
		: ( dynShare >= 8u ? 1u : (uint32_t) ( (unsigned long long) chunks * ( 8u - dynShare ) / ( 8ull * waves ) ) );
	// (one way out: `first` and the answer are wave-uniform values of one definition each)
	uint32_t c = 0xffffffffu;
	const uint32_t base = rounds * waves;
	if ( g.round < rounds )
	{
		c = g.round * waves + wave;
		++g.round;
	}
	else if ( base < chunks )
	{
		uint32_t k = 0;
		if ( lane_id() == 0u )
			k = atomicAdd( head, 1u );
		c = base + wave_bcast( k );
	}
	c = wave_bcast( c );
	first = wave_bcast( c << shift );
	return c < chunks;
}

// Where a wave stands in its reservations: one (next position, end) pair for the hit list of the slot (index 0) and
// one per ray list the pass can append to (index = slot of the list).  They live in LDS, not in registers: they are
// needed for a moment at the end of a chunk, and seven pairs plus the list addresses that follow from them, held
// across the traversal, were a fifth of the kernel's scalar registers (whose overflow goes into vector registers).
__shared__ uint2 g_lds_resv[( RT_BLOCK / 64 ) * ( RT_MAX_SLOTS + 1 )];
DEVI uint2 *resv_slot( int k ) { return g_lds_resv + ( threadIdx.x >> 6 ) * ( RT_MAX_SLOTS + 1 ) + k; }
DEVI Resv resv_load( int k )
{
	const uint2 v = *resv_slot( k );
	Resv r = { wave_bcast( v.x ), wave_bcast( v.y ) };
	return r;
}
DEVI void resv_store( int k, const Resv &r )
{
	if ( lane_id() == 0u )
		*resv_slot( k ) = make_uint2( r.pos, r.end );
	wave_lds_sync();
}
DEVI void resv_init()
{
	if ( lane_id() <= (uint32_t) RT_MAX_SLOTS )
		*resv_slot( (int) lane_id() ) = make_uint2( 0u, 0u );
	wave_lds_sync();
}

DEVI void push_ray( const Work &W, uint32_t block, int target, bool want, const Ray &r, uint32_t pix, int parent, int type, f3 coef, f3 P )
{
	// lanes of one wave may target different slots: one aggregated append per slot that occurs
	unsigned long long todo = __ballot( want );
	while ( todo != 0ull )
	{
		const int s = __builtin_amdgcn_readlane( target, (int) ( __ffsll( (long long) todo ) - 1 ) );
		const bool mine = want && target == s;
#ifdef RT_EXP_REPLAY
		const size_t lbase = (size_t) ( s - 1 ) * W.cap + ( W.replay ? W.altOffset : 0 );
		uint32_t *const counter = W.replay ? &W.ctr->rayCountAlt[s].v : &W.ctr->rayCount[s].v;
#else
		const size_t lbase = (size_t) ( s - 1 ) * W.cap;
		uint32_t *const counter = &W.ctr->rayCount[s].v;
#endif
		Resv rv = resv_load( s );
		uint32_t idx = list_append( counter, rv, mine, block,
			[&]( uint32_t i ) { if ( i < W.cap ) W.rayA[lbase + i].w = __uint_as_float( RT_INVALID ); } );
		resv_store( s, rv );
		if ( mine && idx >= W.cap )
			W.ctr->overflow = 2u;                    // (2 = ray list) reported by the host; the ray is dropped
		else if ( mine )
		{
			size_t o = lbase + idx;
			W.rayA[o] = make_float4( r.src.x, r.src.y, r.src.z, __uint_as_float( pix ) );
			W.rayB[o] = make_float4( r.dest.x, r.dest.y, r.dest.z,
				__uint_as_float( (uint32_t) parent | ( (uint32_t) type << 8 ) ) );
			W.rayC[o] = make_float4( coef.x, coef.y, coef.z, 0.0f );
			W.rayD[o] = make_float4( P.x, P.y, P.z, 0.0f );
		}
		todo &= ~__ballot( mine );
	}
}

// ---------------------------------------------------------------------------
// k_closest: closest-hit pass of one slot
// ---------------------------------------------------------------------------

// what is left of a wave's reservations when it runs out of chunks is marked as holes
DEVI void closest_out_close( const Work &W, int slot )
{
	const size_t hbase = (size_t) slot * W.cap;
	Resv hits = resv_load( 0 );
	list_close( hits, [&]( uint32_t i ) { if ( i < W.cap ) W.hitA[hbase + i].x = __uint_as_float( RT_INVALID ); } );
	for ( int s = 1; s < RT_MAX_SLOTS; ++s )
	{
#ifdef RT_EXP_REPLAY
		const size_t lbase = (size_t) ( s - 1 ) * W.cap + ( W.replay ? W.altOffset : 0 );
#else
		const size_t lbase = (size_t) ( s - 1 ) * W.cap;
#endif
		Resv rays = resv_load( s );
		list_close( rays, [&]( uint32_t i ) { if ( i < W.cap ) W.rayA[lbase + i].w = __uint_as_float( RT_INVALID ); } );
	}
}

// all spheres, CL/rt_funcs_primitives.cl:381-416
struct SphereHit { float t; uint32_t n; int b; };
DEVI SphereHit nearest_sphere( const DevScene &S, const Ray &r )
{
	SphereHit sh = { INFINITY, 0u, 0 };
	for ( int i = 0; i < S.nSpheres; ++i )
	{
		float ts;
		int b = ray_sphere( r, S.spheres[i], ts );
		if ( b && ts < sh.t )
		{
			sh.t = ts;
			sh.n = (uint32_t) i;
			sh.b = b;
		}
	}
	return sh;
}

// What follows the closest-hit query of ray-list entry `e` (CL/rt_funcs_primitives.cl:611-677):
// sphere against triangle, the hit record, the shading normal, the child rays, or
// the finished slot of a miss.  Wave-uniform: every lane of the wave calls it,
// lanes without a ray with valid = false.
DEVI void finish_closest( const DevScene &S, const Work &W, int slot, int wantIds, uint32_t block,
	bool valid, uint32_t e, const RayHead &rh, const NearHit &h, const SphereHit &sph )
{
	const size_t hbase = (size_t) slot * W.cap;
	auto holeHit = [&]( uint32_t i ) { if ( i < W.cap ) W.hitA[hbase + i].x = __uint_as_float( RT_INVALID ); };
	const float prMinT = sph.t;
	const uint32_t prN = sph.n;
	const int prB = sph.b;
	// CL/rt_funcs_primitives.cl:611-633: triangle wins only if strictly closer;
	// "flag set, no valid distance, no sphere" is shaded as a miss
	const bool triHit = valid && ( h.nearestB || prB ) && ( h.minT < prMinT );
	const bool sphHit = valid && ( h.nearestB || prB ) && !( h.minT < prMinT ) && prB;
	bool hit = triHit || sphHit;

	Resv hits = resv_load( 0 );
	const uint32_t hidx = list_append( &W.ctr->hitCount[slot].v, hits, hit, block, holeHit );
	resv_store( 0, hits );
	if ( hit && hidx >= W.cap )
	{
		W.ctr->overflow = 1u;                        // (1 = hit list) reported by the host; the entry is dropped
		hit = false;
		valid = false;
	}
	bool pushRefl = false, pushRefr = false;
	Ray rRefl, rRefr;
	f3 P = F3( 0.0f, 0.0f, 0.0f ), cRefl = P, cRefr = P;
	int sp = 0;
	if ( hit )
	{
		const size_t ho = hbase + hidx;
		const float tHit = triHit ? h.minT : prMinT;
		const uint32_t id = triHit ? h.prN : ( prN | 0x80000000u );
		const int bNearest = triHit ? h.nearestB : prB;
		W.hitA[ho] = make_float4( __uint_as_float( e ), tHit, h.minU, h.minV );
		W.hitB[ho] = make_uint2( id, (uint32_t) bNearest );

		// shading normal and child rays of the hit (the next slot's closest-hit pass
		// waits for nothing else)
		f3 N;
		uint32_t matIdx;
		// hit point: dest * t + src for triangles (:66) and spheres (:42-43) alike
		P = add3( mul3( rh.ray.dest, tHit ), rh.ray.src );
		if ( id & 0x80000000u )
		{
			// sphere normal, CL/rt_funcs_primitives.cl:45-46
			const uint32_t si = id & 0x7fffffffu;
			const float4 sp4 = S.spheres[si];
			matIdx = S.sphereMat[si];
			N = div3( sub3( P, F3( sp4.x, sp4.y, sp4.z ) ), sp4.w );
			N = mul3( N, (float) bNearest );
		}
		else
		{
			// barycentric normal (not normalised), :68-71
			const float4 *sp4 = S.shade + 3u * (size_t) id;
			const float4 a = sp4[0], b = sp4[1], c = sp4[2];
			const f3 n0 = F3( a.x, a.y, a.z ), n1 = F3( a.w, b.x, b.y ), n2 = F3( b.z, b.w, c.x );
			const float u = h.minU, v = h.minV;
			matIdx = __float_as_uint( c.y );
			N = add3( add3( mul3( n1, u ), mul3( n2, v ) ), mul3( n0, 1.0f - u - v ) );
		}
		W.hitN[ho] = make_float4( N.x, N.y, N.z, __uint_as_float( matIdx ) );

		const rt_material &mat = S.mats[matIdx];
		const f3 refl = rgb_of( mat.reflect );
		const float alpha = mat.color.a;

		sp = slot == 0 ? 0 : (int) W.stackPos[rh.pix];
#ifdef RT_EXP_REPLAY
		if ( W.replay ) sp = slot;          // (a chain of reflections: the pixel's last slot when this one was first processed)
#endif

		// reflection ray :643-655
		if ( ( ( sp + 1 ) < S.stackSize ) && ( bNearest != -1 ) && ( ( refl.x + refl.y + refl.z ) > 0.0f ) )
		{
			++sp;
			pushRefl = true;
			rRefl.dest = reflect3( rh.ray.dest, N );
			rRefl.src = add3( P, mul3( rRefl.dest, EPSILON ) );
			cRefl = refl;
		}
		// refraction ray :657-677
		if ( ( ( sp + 1 ) < S.stackSize ) && ( alpha < 0.99f ) )
		{
			const float nRel = ENV_OPT_DENSITY / mat.optDens;
			const float cosI = dot3( N, rh.ray.dest );
			const float sinT = nRel * nRel * ( 1.0f - cosI * cosI );
			if ( sinT < 1.0f )
			{
				++sp;
				pushRefr = true;
				rRefr.dest = norm3( sub3( mul3( rh.ray.dest, nRel ),
					mul3( N, nRel * cosI + sqrtf( 1.0f - sinT ) ) ) );
				rRefr.src = add3( P, mul3( rRefr.dest, EPSILON ) );
				const float lf = mat.lightFalloff;
				cRefr = F3( mat.color.r * lf, mat.color.g * lf, mat.color.b * lf );
			}
		}
#ifdef RT_EXP_REPLAY
		if ( !W.replay )
#endif
		if ( slot == 0 || pushRefl || pushRefr )
			W.stackPos[rh.pix] = (uint32_t) sp;
	}
	// slot numbers: reflection takes the first new slot, refraction the next
	push_ray( W, block, sp - ( pushRefr ? 1 : 0 ), pushRefl, rRefl, rh.pix, slot, 1, cRefl, P );
	push_ray( W, block, sp, pushRefr, rRefr, rh.pix, slot, 0, cRefr, P );
	if ( !hit && valid )
	{
		const size_t o = (size_t) slot * W.npix + rh.pix;
		W.res[o] = make_float4( 0.0f, 0.0f, 0.0f, __int_as_float( parent_of( rh.pk ) ) );
		if ( slot == 0 )
		{
#ifdef RT_EXP_REPLAY
			if ( !W.replay )
#endif
			W.stackPos[rh.pix] = 0u;
		}
		else
		{
			f3 fw = fold_weight( W, slot, e, rh.pk, F3( 0.0f, 0.0f, 0.0f ) );
			W.fac[o] = make_float4( fw.x, fw.y, fw.z, 0.0f );
		}
	}
	if ( slot == 0 && wantIds && valid )
		W.hitId[rh.pix] = triHit ? (int32_t) h.prN : ( sphHit ? -2 - (int32_t) prN : -1 );
}

// number of entries of ray list `slot` (reservation holes included); never beyond the buffer
DEVI uint32_t ray_list_count( const Work &W, int slot )
{
	if ( slot == 0 )
		return W.nChunks0 * 64u;
	const uint32_t n = W.ctr->rayCount[slot].v;
	return n < W.cap ? n : W.cap;
}
DEVI uint32_t hit_list_count( const Work &W, int slot )
{
	const uint32_t n = W.ctr->hitCount[slot].v;
	return n < W.cap ? n : W.cap;
}

// rays a wave takes per chunk of a list of `count` entries, as a power of two:
// 64 (one per lane) when that keeps every resident wave busy, fewer for short
// lists, so that the rays are spread over the chip and walk cooperatively
// from the start with the lanes that would otherwise idle
DEVI uint32_t chunk_shift( uint32_t count, uint32_t coopLive )
{
	uint32_t shift = 6u;
	if ( coopLive != 0u )
	{
		// (bits 8..15 of coopLive: development knob, waves wanted per resident wave in eighths)
		const uint32_t waves = gridDim.x * ( RT_BLOCK / 64u ) * ( ( coopLive >> 8 ) & 0xffu ) / 8u;
		while ( shift > RT_COOP_MIN_SHIFT && ( ( count + ( 1u << shift ) - 1u ) >> shift ) < waves ) --shift;
	}
	return shift;
}

template <bool STATS, int WAVES>
__global__ __launch_bounds__( RT_BLOCK, WAVES ) void k_closest( DevScene S, Work W, int slot, int wantIds )
{
	const StackRef st = make_stack( W );
	Tally tl = { 0u, 0u, 0u, 0u, 0.0f, 0.0f };

	const uint32_t count = ray_list_count( W, slot );
	const uint32_t shift = chunk_shift( count, W.coopLive );
	uint32_t first;
	Grab grab = { 0u };
	const uint32_t block = resv_block( W, count );
	resv_init();
	while ( next_chunk( &W.ctr->headClosest[slot].v, count, grab, first, RT_TRAV_DYN, shift ) )
	{
#ifdef RT_PROFILE_CHUNKS
		const unsigned long long cs = wall_clock64();
#endif
		const uint32_t e = first + lane_id();
		RayHead rh;
		rh.pix = rh.pk = 0u;
		const bool valid = lane_id() < ( 1u << shift ) && load_ray<true>( S, W, slot, e, count, rh );

		SphereHit sph = { INFINITY, 0u, 0 };
		NearHit h = { INFINITY, 0.0f, 0.0f, 0u, 0 };
		Trav t;
		bool live = false;

		if ( valid )
		{
			sph = nearest_sphere( S, rh.ray );
			// Neighbouring lanes carry neighbouring rays and walk the tree in step,
			// so their node and record loads fall into the same cache lines.
			// (Refilling finished lanes with fresh rays mid-flight was measured and
			// halves the speed on this workload: it destroys exactly that locality.)
			live = trav_begin<STATS>( S, rh.ray, t, tl );
		}
#ifdef RT_PROFILE_CHUNKS
		Prof pf = { 0u, 0u, 0u, 0ull, 0ull, 0ull, 0ull, 0ull };
		{
			const uint32_t oct = ( rh.ray.dest.x < 0.0f ? 1u : 0u ) | ( rh.ray.dest.y < 0.0f ? 2u : 0u ) | ( rh.ray.dest.z < 0.0f ? 4u : 0u );
			uint32_t distinct = 0u, major = 0u;
			for ( uint32_t o = 0u; o < 8u; ++o )
			{
				const uint32_t n = (uint32_t) __popcll( __ballot( valid && oct == o ) );
				distinct += n ? 1u : 0u;
				major = n > major ? n : major;
			}
			const unsigned long long nValid = (unsigned long long) __popcll( __ballot( valid ) );
			if ( lane_id() == 0u )
			{
				atomicAdd( &W.ctr->octDistinct[slot], (unsigned long long) distinct );
				atomicAdd( &W.ctr->octMajor[slot], (unsigned long long) major );
				atomicAdd( &W.ctr->octRays[slot], nValid );
			}
		}
		const unsigned long long c0 = wall_clock64();
#endif
		nearest_wave<STATS>( S, rh.ray, tri_ray( rh.ray.src, rh.ray.dest ), st, live, t, h, tl, W.coopLive & 0xffu, W.coopCap, shift < 6u && ( W.coopLive >> 31 ) PROF_PASS );
#ifdef RT_PROFILE_CHUNKS
		const unsigned long long c1 = wall_clock64();
#endif
		finish_closest( S, W, slot, wantIds, block, valid, e, rh, h, sph );
#ifdef RT_PROFILE_CHUNKS
		if ( lane_id() == 0u )
		{
			atomicAdd( &W.ctr->tFinish[slot], wall_clock64() - c1 );
			atomicAdd( &W.ctr->tSetup[slot], c0 - cs );
			atomicAdd( &W.ctr->tDescend[slot], pf.tDescend );
			atomicAdd( &W.ctr->tLeaf[slot], pf.tLeaf );
			const unsigned long long dt = ( wall_clock64() - c0 ) / 1000ull;       // 100 MHz ticks -> 10 us
			atomicAdd( &W.ctr->chunkHist[slot][dt < 63ull ? dt : 63ull], 1u );
			atomicAdd( &W.ctr->lockRounds[slot], (unsigned long long) pf.lock );
			atomicAdd( &W.ctr->coopRounds[slot], (unsigned long long) pf.coop );
			atomicAdd( &W.ctr->coopIters[slot], (unsigned long long) pf.iters );
			atomicAdd( &W.ctr->chunksSeen[slot], 1ull );
			atomicAdd( &W.ctr->tExpand[slot], pf.tExpand );
			atomicAdd( &W.ctr->tScan[slot], pf.tScan );
			atomicAdd( &W.ctr->tCoop[slot], pf.tCoop );
			atomicAdd( &W.ctr->tChunk[slot], wall_clock64() - c0 );
			atomicMax( &W.ctr->maxLock[slot], pf.lock );
			atomicMax( &W.ctr->maxCoop[slot], pf.coop );
			atomicMax( &W.ctr->maxIters[slot], pf.iters );
		}
#endif
	}
	closest_out_close( W, slot );
	if ( STATS )
		flush_tally( W, tl, false );
}

// ---------------------------------------------------------------------------
// k_shadow: one shadow query per (hit, light) of one slot
// ---------------------------------------------------------------------------

// light record: 3 x float4 = (col b,g,r,a) (pos.xyz,_) (rad,_,_,_)
DEVI void to_light( const DevScene &S, int o, f3 P, f3 &toLight, float &d )
{
	float4 lp = S.lights[3 * o + 1];
	toLight = sub3( F3( lp.x, lp.y, lp.z ), P );
	d = len3( toLight );
	toLight = div3( toLight, d );
}

// Item i = light (i / hits) of hit (i % hits).  Stores the two clamped opacity
// sums of CL/rt_funcs_primitives.cl:481-502 added together; the variable that
// carries the sphere sum keeps it when the ray misses the scene box, so the
// sphere opacity is then counted twice, as in the reference.
template <bool STATS>
__global__ __launch_bounds__( RT_BLOCK, STATS ? RT_WAVES_SHADOW_STATS : RT_WAVES_SHADOW ) void k_shadow( DevScene S, Work W, int slot )
{
	const StackRef st = make_stack( W );
	Tally tl = { 0u, 0u, 0u, 0u, 0.0f, 0.0f };

	const uint32_t hits = hit_list_count( W, slot );
	const uint32_t rayCount = ray_list_count( W, slot );
	const uint32_t total = hits * (uint32_t) S.nLights;      // < 2^31, checked by the host
	const uint32_t shift = chunk_shift( total, W.coopLive );
	uint32_t first;
	Grab grab = { 0u };
	while ( next_chunk( &W.ctr->headShadow[slot].v, total, grab, first, RT_TRAV_DYN, shift ) )
	{
		const uint32_t i = first + lane_id();
		bool item = lane_id() < ( 1u << shift ) && i < total;
		uint32_t o = 0u, hh = 0u;
		float4 ha = make_float4( 0.0f, 0.0f, 0.0f, 0.0f );
		RayHead rh;
		if ( item )
		{
			o = i / hits;
			hh = i - o * hits;
			ha = W.hitA[(size_t) slot * W.cap + hh];
			item = __float_as_uint( ha.x ) != RT_INVALID && load_ray( S, W, slot, __float_as_uint( ha.x ), rayCount, rh );
		}

		Ray sh;
		Trav t;
		float d = 0.0f, a = 0.0f, shadowed = 0.0f;
		bool live = false;
		if ( item )
		{
			const f3 P = add3( mul3( rh.ray.dest, ha.y ), rh.ray.src );
			f3 toLight;
			to_light( S, (int) o, P, toLight, d );
			sh.dest = toLight;
			sh.src = add3( P, mul3( toLight, EPSILON ) );
			sh.inv = rcp3( toLight );
			for ( int k = 0; k < S.nSpheres; ++k )           // :418-448
			{
				float ts;
				int b = ray_sphere( sh, S.spheres[k], ts );
				if ( b && ts < d )
					a += S.mats[S.sphereMat[k]].color.a;
			}
			shadowed = clampf( a, 0.0f, 1.0f );
			// :289-293: the triangle sum starts at zero only once the scene box is hit
			if ( trav_begin<STATS>( S, sh, t, tl ) )
			{
				a = 0.0f;
				// Only hits with tNear < t < tFar of their leaf AND t < d count, so the
				// part of the ray beyond the light cannot contribute: walk
				// [tNear, min(tFar, d)].  Leaves keep the same split distances, hence
				// the same hits are counted.
				live = true;
				if ( !axis_parallel( sh ) )
				{
					if ( d < t.tFar )
						t.tFar = d;
					live = t.tNear < t.tFar;
				}
			}
		}
#ifdef RT_PROFILE_CHUNKS
		Prof pf = { 0u, 0u, 0u, 0ull, 0ull, 0ull, 0ull, 0ull };
#endif
		alpha_wave<STATS>( S, sh, tri_ray( sh.src, sh.dest ), st, live, t, d, a, tl, W.coopLive & 0xffu, W.coopCap, shift < 6u && ( W.coopLive >> 31 ) PROF_PASS );
		if ( item )
			W.shadow[( (size_t) slot * S.nLights + o ) * W.cap + hh] = shadowed + clampf( a, 0.0f, 1.0f );
	}
	if ( STATS )
		flush_tally( W, tl, true );
}

// ---------------------------------------------------------------------------
// lighting of the compacted hits of one slot (k_light)
// ---------------------------------------------------------------------------

// CL/rt_funcs_primitives.cl:450-564 with the shadow term taken from k_shadow
DEVI f3 light_point( const DevScene &S, const Work &W, int slot, uint32_t h, f3 P, f3 N, const rt_material &mat, f3 viewer )
{
	f3 col = F3( 0.0f, 0.0f, 0.0f );
	for ( int o = 0; o < S.nLights; ++o )
	{
		float4 lc = S.lights[3 * o], lr = S.lights[3 * o + 2];
		f3 lcol = F3( lc.z, lc.y, lc.x );          // stored b,g,r,a
		float rad = lr.x;

		f3 toLight;
		float d;
		to_light( S, o, P, toLight, d );

		float shadowed = W.shadow[( (size_t) slot * S.nLights + o ) * W.cap + h];
		shadowed *= mat.color.a;

		f3 ambient = col_mult( lcol, rgb_of( mat.ambient ) );

		float ang = cl_max( 0.0f, dot3( N, toLight ) );
		f3 diffuse = col_mult( lcol, rgb_of( mat.diffuse ) );
		diffuse = col_mult( diffuse, rgb_of( mat.color ) );
		diffuse = col_scale( diffuse, ang * rad / d );

		toLight = mul3( toLight, -1.0f );
		toLight = reflect3( toLight, N );
		f3 toEye = norm3( sub3( viewer, P ) );
		ang = dot3( toEye, toLight );
		float specFact = cl_pow( cl_max( ang, 0.0f ), mat.specular.a );
		f3 specular = col_mult( rgb_of( mat.specular ), lcol );
		specular = col_scale( specular, specFact * rad / d );

		diffuse = col_add( ambient, diffuse );
		f3 tmp = col_add( diffuse, specular );
		tmp = col_scale( tmp, 1.0f - shadowed );
		col = col_add( tmp, col );
	}
	return col;
}

// k_light: colour of every hit of one slot (needs k_shadow's sums and
// k_closest's normals) and the weight of the slot in its parent
__global__ __launch_bounds__( RT_BLOCK ) void k_light( DevScene S, Work W, int slot )
{
	const uint32_t count = hit_list_count( W, slot );
	const uint32_t rayCount = ray_list_count( W, slot );
	const size_t base = (size_t) slot * W.cap;
	uint32_t first;
	Grab grab = { 0u };
	while ( next_chunk( &W.ctr->headShade[slot].v, count, grab, first, RT_STREAM_DYN ) )
	{
		const uint32_t h = first + lane_id();
		if ( h >= count )
			continue;
		const float4 ha = W.hitA[base + h];
		RayHead rh;
		if ( __float_as_uint( ha.x ) == RT_INVALID || !load_ray( S, W, slot, __float_as_uint( ha.x ), rayCount, rh ) )
			continue;
		const f3 P = add3( mul3( rh.ray.dest, ha.y ), rh.ray.src );
		const float4 hn = W.hitN[base + h];
		const rt_material mat = S.mats[__float_as_uint( hn.w )];
		const f3 col = light_point( S, W, slot, h, P, F3( hn.x, hn.y, hn.z ), mat, rh.ray.src );

		const size_t o = (size_t) slot * W.npix + rh.pix;
		W.res[o] = make_float4( col.x, col.y, col.z, __int_as_float( parent_of( rh.pk ) ) );
		if ( slot > 0 )
		{
			const f3 fw = fold_weight( W, slot, __float_as_uint( ha.x ), rh.pk, P );
			W.fac[o] = make_float4( fw.x, fw.y, fw.z, 0.0f );
		}
	}
}

// ---------------------------------------------------------------------------
// k_fold: children into parents, last slot first (CL/rt_funcs_primitives.cl:688-725),
// then the truncating 8-bit store (CL/rt_raytrace.cl:83-86)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__( RT_BLOCK ) void k_fold( Work W, rt_argb *out, int stackSize )
{
	uint32_t p = blockIdx.x * RT_BLOCK + threadIdx.x;
	if ( p >= W.npix )
		return;
	int n = (int) W.stackPos[p] + 1;
	f3 col[RT_MAX_SLOTS];
	f3 fw[RT_MAX_SLOTS];
	int par[RT_MAX_SLOTS];
#pragma unroll
	for ( int s = 0; s < RT_MAX_SLOTS; ++s )
	{
		col[s] = fw[s] = F3( 0.0f, 0.0f, 0.0f );
		par[s] = -1;
		if ( s < n && s < stackSize )
		{
			float4 r = W.res[(size_t) s * W.npix + p];
			col[s] = F3( r.x, r.y, r.z );
			par[s] = __float_as_int( r.w );
			if ( s > 0 )
			{
				float4 f = W.fac[(size_t) s * W.npix + p];
				fw[s] = F3( f.x, f.y, f.z );
			}
		}
	}
#pragma unroll
	for ( int c = RT_MAX_SLOTS - 1; c >= 1; --c )
	{
		if ( c < n )
		{
			f3 add = F3( col[c].x * fw[c].x, col[c].y * fw[c].y, col[c].z * fw[c].z );
#pragma unroll
			for ( int q = 0; q < c; ++q )
				if ( par[c] == q )
					col[q] = clamp01( add3( col[q], add ) );
		}
	}
	rt_argb px;
	px.a = (unsigned char) ( 255.0f * 0.0f );
	px.r = (unsigned char) ( 255.0f * col[0].x );
	px.g = (unsigned char) ( 255.0f * col[0].y );
	px.b = (unsigned char) ( 255.0f * col[0].z );
	out[p] = px;
}

// the camera matrices of the frame into device memory (DevScene::cam)
__global__ __launch_bounds__( 64 ) void k_put_camera( DevScene S, float *dst )
{
	const uint32_t i = threadIdx.x;
	if ( i < 16u ) dst[i] = S.world[i];
	else if ( i < 32u ) dst[i] = S.vtp[i - 16u];
}

// number of entries of a list that are not reservation holes (statistics only);
// word = index of the 32-bit word of the float4 that carries the marker
__global__ __launch_bounds__( RT_BLOCK ) void k_count_valid( const float4 *list, const uint32_t *n, int word, unsigned long long *out )
{
	uint32_t cnt = 0;
	for ( uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x; i < *n; i += gridDim.x * RT_BLOCK )
		cnt += ( (const uint32_t *) ( list + i ) )[word] != RT_INVALID;
	const unsigned long long m = __ballot( cnt != 0u );
	for ( int o = 32; o > 0; o >>= 1 )
		cnt += __shfl_down( cnt, o );
	if ( lane_id() == 0u && m )
		atomicAdd( out, (unsigned long long) cnt );
}

// every float bit pattern: rcp_exact against the IEEE division
__global__ __launch_bounds__( RT_BLOCK ) void k_selftest_rcp( unsigned long long *mismatches, unsigned long long *fastLanes )
{
	unsigned long long bad = 0, fast = 0;
	for ( unsigned long long i = (unsigned long long) blockIdx.x * RT_BLOCK + threadIdx.x; i < ( 1ull << 32 ); i += (unsigned long long) gridDim.x * RT_BLOCK )
	{
		const float x = __uint_as_float( (uint32_t) i );
		bool ok;
		const float n = rcp_newton( x, ok );
		const float q = 1.0f / x;
		fast += ok;
		// the value the kernels use: Newton where the estimate is a normal number, the division elsewhere
		const float used = ok ? n : q;
		bad += __float_as_uint( used ) != __float_as_uint( q ) && !( q != q && used != used );
	}
	atomicAdd( mismatches, bad );
	atomicAdd( fastLanes, fast );
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

static thread_local char g_err[512] = "";

static int fail( const char *fmt, ... )
{
	va_list ap;
	va_start( ap, fmt );
	vsnprintf( g_err, sizeof( g_err ), fmt, ap );
	va_end( ap );
	return 1;
}

#define HIPCHK( x ) do { hipError_t e_ = ( x ); if ( e_ != hipSuccess ) \
	return fail( "%s failed: %s (%s:%d)", #x, hipGetErrorString( e_ ), __FILE__, __LINE__ ); } while ( 0 )

struct DevBuf
{
	void *p = nullptr;
	size_t cap = 0;
};

// What a context that lends its scene (rt_hip_share_scene) and its borrowers have in common:
// a generation number that every upload into the lender - and its destruction - advances, so
// that a borrower notices that the device pointers it copied may no longer be the lender's.
struct SceneTag
{
	std::atomic<uint64_t> gen{ 1 };
	std::atomic<int> refs{ 1 };
};
static void scene_tag_release( SceneTag *t )
{
	if ( t && t->refs.fetch_sub( 1 ) == 1 )
		delete t;
}

struct rt_hip_ctx
{
	int device = 0;
	int numCU = 0;
	hipStream_t own = nullptr, stream = nullptr;
	DevScene S;
	Work W;
	DevBuf nodes, recs, shade, mats, spheres, sphereMat, lights, leafFull;
	DevBuf ctr, rays, hitA, hitB, hitN, res, fac, shadow, stackPos, hitId, spill, frame;
	DevBuf cam;                   // camera matrices of the frame being rendered (allocated with the context)
	uint64_t nNodes = 0, nRecs = 0, nTris = 0, nMats = 0;
	uint64_t maxMatTree = 0, maxMatSphere = 0;   // highest material slot referenced
	int treeDepth = 0;
	int allocStack = 0;           // stack size the work buffers were sized for
	bool workDirty = true;
	int gridClosest[2][2] = { { 0, 0 }, { 0, 0 } };   // [big frame][stats]
	int gridShadow[2] = { 0, 0 }, gridLight = 0;
	int coopLive = RT_COOP_LIVE, coopShadow = 1, coopClosest = 1, coopCap = RT_LDS_STACK, coopFill = 4, coopAtOnce = 0;
	uint32_t bigFrame = RT_BIG_FRAME;
	int occShadow[2] = { 1, 1 };
	hipStream_t aux[RT_AUX_STREAMS] = {};        // shadow + lighting passes (slot k on stream k mod RT_AUX_STREAMS)
	hipEvent_t evHit[RT_MAX_SLOTS] = {}, evLit[RT_MAX_SLOTS] = {};
	int overlap = 1;
	int workLights = -1;          // light count the shadow buffer was sized for
	int maxGrid = 0;
	int launchWaves = RT_OCC_CAP;                     // RT_PARAM_LAUNCH_WAVES
	int pruneLeaves = 0;                              // RT_PARAM_PRUNE_LEAVES (trees built by rt_hip_build_tree)
	uint64_t stageTicket = 0;                         // geometry copied ahead of the next rt_hip_build_tree (rt_hip_stage_geometry)
	// HIP events of the last RT_TIMED_FRAMES frames rendered with RT_HIP_TIMING
	// [frame][(slot * 3 + kernel) * 2 + {start, stop}], then fold start / stop, frame start / stop; of the
	// start events only that of the first kernel is recorded (enqueue_frame, frame_times)
	hipEvent_t ev[RT_TIMED_FRAMES][6 * RT_MAX_SLOTS + 4] = {};
	bool evOverlap[RT_TIMED_FRAMES] = {};   // schedule of each of them (decides which kernel waited for which)
	bool haveEvents = false;
	int timedSlots = 0, timedFrames = 0;
	uint32_t validPixels = 0;
	Counters *lastCtr = nullptr;  // counters of the frame enqueued last (either entry point): where its overflow flag is read
	uint32_t lastCap = 0;
#ifdef RT_EXP_REPLAY
	int replay = 0;               // set_param key 100
	hipEvent_t evRepStart = nullptr, evRepDone[RT_MAX_SLOTS] = {};
#endif
	SceneTag *tag = nullptr;      // generation of the scene this context holds (and may lend)
	SceneTag *lender = nullptr;   // rt_hip_share_scene: the lender's tag and its generation at that time
	uint64_t lenderGen = 0;
};

// every change of a context's scene buffers (upload, adoption of a tree)
static void scene_changed( rt_hip_ctx *c ) { if ( c->tag ) c->tag->gen.fetch_add( 1 ); }
// a borrower may only render while the lender's scene is the one it copied the pointers of
static int lender_unchanged( const rt_hip_ctx *c, const char *who )
{
	if ( c->lender && c->lender->gen.load() != c->lenderGen )
		return fail( "%s: the context whose scene this one shares has uploaded a scene again or was destroyed; call rt_hip_share_scene again", who );
	return 0;
}

const char *rt_hip_last_error( void ) { return g_err; }

int rt_hip_device_count( void )
{
	int n = 0;
	if ( hipGetDeviceCount( &n ) != hipSuccess )
		return 0;
	return n;
}

static int ensure( rt_hip_ctx *c, DevBuf &b, size_t bytes )
{
	if ( bytes == 0 ) bytes = 256;
	if ( b.cap >= bytes )
		return 0;
	if ( b.p )
	{
		HIPCHK( hipStreamSynchronize( c->stream ) );
		HIPCHK( hipFree( b.p ) );
		b.p = nullptr;
		b.cap = 0;
	}
	HIPCHK( hipMalloc( &b.p, bytes ) );
	b.cap = bytes;
	return 0;
}

static int upload( rt_hip_ctx *c, DevBuf &b, const void *src, size_t bytes )
{
	if ( ensure( c, b, bytes ) )
		return 1;
	if ( bytes )
	{
		// pageable source: the copy is staged before the call returns
		HIPCHK( hipMemcpyAsync( b.p, src, bytes, hipMemcpyHostToDevice, c->stream ) );
		HIPCHK( hipStreamSynchronize( c->stream ) );
	}
	return 0;
}

void rt_hip_destroy( rt_hip_ctx *c );
static int scratch_lookup( int device );      // device address of the allocation-free entry point's work area

// Streams come from a pool per device and flavour and go back to it; none is destroyed before the process ends.  The
// runtime deals its hardware queues (GPU_MAX_HW_QUEUES of them) to streams in the order the streams are created,
// whether earlier ones still exist or not: a process that opens and closes contexts would sooner or later find the
// streams of two contexts that run side by side on one queue, where the frames of one wait for the other's (measured:
// a workload rendered after others had come and gone, 41 ms per frame against 27 ms).  With the pool a process owns
// as many streams as it ever used at one time.
#define RT_MAX_DEVICES 64          // device ordinals the library keeps per-device state for
namespace {
struct StreamPool
{
	std::mutex m;
	std::vector<hipStream_t> idle[RT_MAX_DEVICES][2];      // [device][0: blocking flavour (hipStreamCreate), 1: non-blocking]
};
StreamPool g_streams;

hipStream_t stream_take( int device, bool nonBlocking )
{
	if ( device >= 0 && device < RT_MAX_DEVICES )
	{
		std::lock_guard<std::mutex> lk( g_streams.m );
		auto &v = g_streams.idle[device][nonBlocking ? 1 : 0];
		if ( !v.empty() )
		{
			hipStream_t st = v.back();
			v.pop_back();
			return st;
		}
	}
	hipStream_t st = nullptr;
	const hipError_t e = nonBlocking ? hipStreamCreateWithFlags( &st, hipStreamNonBlocking ) : hipStreamCreate( &st );
	return e == hipSuccess ? st : nullptr;
}

void stream_give( int device, bool nonBlocking, hipStream_t st )
{
	if ( !st )
		return;
	(void) hipStreamSynchronize( st );
	if ( device < 0 || device >= RT_MAX_DEVICES )
	{
		(void) hipStreamDestroy( st );
		return;
	}
	std::lock_guard<std::mutex> lk( g_streams.m );
	g_streams.idle[device][nonBlocking ? 1 : 0].push_back( st );
}
} // namespace

rt_hip_ctx *rt_hip_create( int device )
{
	int n = rt_hip_device_count();
	if ( device < 0 || device >= n )
	{
		fail( "rt_hip_create: device %d not available (%d HIP devices)", device, n );
		return nullptr;
	}
	if ( hipSetDevice( device ) != hipSuccess )
	{
		fail( "hipSetDevice(%d) failed", device );
		return nullptr;
	}
	rt_hip_ctx *c = new rt_hip_ctx();
	c->device = device;
	c->tag = new SceneTag();
	hipDeviceProp_t prop;
	if ( hipGetDeviceProperties( &prop, device ) != hipSuccess || ( c->own = stream_take( device, false ) ) == nullptr )
	{
		fail( "rt_hip_create: device query / stream creation failed" );
		scene_tag_release( c->tag );
		delete c;
		return nullptr;
	}
	c->numCU = prop.multiProcessorCount;
	c->stream = c->own;
	// this GPU's address of the allocation-free entry point's work area
	if ( scratch_lookup( device ) )
	{
		rt_hip_destroy( c );
		return nullptr;
	}
	memset( &c->S, 0, sizeof( c->S ) );
	memset( &c->W, 0, sizeof( c->W ) );
	if ( ensure( c, c->cam, 32 * sizeof( float ) ) )
	{
		rt_hip_destroy( c );
		return nullptr;
	}
	c->S.cam = (const float *) c->cam.p;
	c->S.stackSize = STACK_SIZE;
	c->S.bandH = 16;
	c->S.nParts = 1;
	c->S.part = 0;
	for ( int i = 0; i < 16; ++i )
		c->S.world[i] = c->S.vtp[i] = ( i % 5 == 0 ) ? 1.0f : 0.0f;

	// persistent grids: one resident wave set per CU, as many blocks as fit
	int occ = 0;
	struct { const void *f; int *dst; int *occ; } ks[7] = {
		{ (const void *) k_closest<false, RT_WAVES_CLOSEST>, &c->gridClosest[0][0], nullptr }, { (const void *) k_closest<true, RT_WAVES_CLOSEST_STATS>, &c->gridClosest[0][1], nullptr },
		{ (const void *) k_closest<false, RT_WAVES_CLOSEST_BIG>, &c->gridClosest[1][0], nullptr }, { (const void *) k_closest<true, RT_WAVES_CLOSEST_STATS>, &c->gridClosest[1][1], nullptr },
		{ (const void *) k_shadow<false>, &c->gridShadow[0], &c->occShadow[0] }, { (const void *) k_shadow<true>, &c->gridShadow[1], &c->occShadow[1] },
		{ (const void *) k_light, &c->gridLight, nullptr } };
	for ( auto &k : ks )
	{
		if ( hipOccupancyMaxActiveBlocksPerMultiprocessor( &occ, k.f, RT_BLOCK, 0 ) != hipSuccess || occ < 1 )
			occ = 1;
		if ( occ * ( RT_BLOCK / 64 ) > RT_MAX_LAUNCH_WAVES ) occ = RT_MAX_LAUNCH_WAVES / ( RT_BLOCK / 64 );
		*k.dst = occ * c->numCU;
		if ( k.occ ) *k.occ = occ;
		if ( *k.dst > c->maxGrid ) c->maxGrid = *k.dst;
	}
	// development knob
	if ( const char *v = getenv( "RT_COOP_LIVE" ) ) c->coopLive = atoi( v );
	if ( c->coopLive < 0 || c->coopLive > 32 ) c->coopLive = 0;
	if ( const char *v = getenv( "RT_COOP_FILL" ) ) c->coopFill = atoi( v ) > 0 && atoi( v ) < 256 ? atoi( v ) : 4;
	if ( const char *v = getenv( "RT_COOP_ATONCE" ) ) c->coopAtOnce = atoi( v );
	if ( const char *v = getenv( "RT_COOP_SHADOW" ) ) c->coopShadow = atoi( v );
	if ( const char *v = getenv( "RT_COOP_CLOSEST" ) ) c->coopClosest = atoi( v );
	if ( const char *v = getenv( "RT_BIG_FRAME" ) ) c->bigFrame = (uint32_t) atoll( v );
	if ( const char *v = getenv( "RT_LAUNCH_WAVES" ) ) if ( atoi( v ) >= 1 && atoi( v ) <= RT_MAX_LAUNCH_WAVES ) c->launchWaves = atoi( v );
	// a streaming pass gains nothing from more than a few waves per SIMD
	if ( c->gridLight > RT_STREAM_OCC * c->numCU ) c->gridLight = RT_STREAM_OCC * c->numCU;

	// Ordinary priority: low-priority auxiliary streams bought nothing once their grids were
	// limited, and when two processes shared one GPU their queues were starved for good
	// (a frame never completed in about one run out of three).
	bool ok = true;
	for ( auto &a : c->aux )
		ok = ok && ( a = stream_take( device, true ) ) != nullptr;
	for ( int k = 0; k < RT_MAX_SLOTS && ok; ++k )
		ok = hipEventCreateWithFlags( &c->evHit[k], hipEventDisableTiming ) == hipSuccess
			&& hipEventCreateWithFlags( &c->evLit[k], hipEventDisableTiming ) == hipSuccess;
	if ( !ok )
	{
		fail( "rt_hip_create: auxiliary stream / event creation failed" );
		rt_hip_destroy( c );
		return nullptr;
	}
	return c;
}

void rt_hip_destroy( rt_hip_ctx *c )
{
	if ( !c )
		return;
	(void) hipSetDevice( c->device );
	(void) hipStreamSynchronize( c->stream );
	for ( auto &a : c->aux ) stream_give( c->device, true, a );
	for ( int k = 0; k < RT_MAX_SLOTS; ++k )
	{
		if ( c->evHit[k] ) (void) hipEventDestroy( c->evHit[k] );
		if ( c->evLit[k] ) (void) hipEventDestroy( c->evLit[k] );
	}
	DevBuf *all[] = { &c->nodes, &c->recs, &c->shade, &c->mats, &c->spheres, &c->sphereMat, &c->lights, &c->cam, &c->leafFull,
		&c->ctr, &c->rays, &c->hitA, &c->hitB, &c->hitN, &c->res, &c->fac, &c->shadow, &c->stackPos, &c->hitId, &c->spill, &c->frame };
	for ( DevBuf *b : all )
		if ( b->p ) (void) hipFree( b->p );
	if ( c->haveEvents )
		for ( auto &f : c->ev ) for ( auto &e : f ) (void) hipEventDestroy( e );
	stream_give( c->device, false, c->own );
	if ( c->tag ) c->tag->gen.store( ~0ull );        // borrowers: the scene is gone
	scene_tag_release( c->tag );
	scene_tag_release( c->lender );
	delete c;
}

// A context that runs on the caller's stream gives its own stream back to the pool: every stream of the process
// takes a place among the runtime's hardware queues, and streams beyond their number share queues with others.
int rt_hip_set_stream( rt_hip_ctx *c, void *s )
{
	if ( !c ) return fail( "null context" );
	HIPCHK( hipSetDevice( c->device ) );
	HIPCHK( hipStreamSynchronize( c->stream ) );
	if ( s )
	{
		stream_give( c->device, false, c->own );
		c->own = nullptr;
		c->stream = (hipStream_t) s;
		return 0;
	}
	if ( !c->own && ( c->own = stream_take( c->device, false ) ) == nullptr )
		return fail( "rt_hip_set_stream: stream creation failed" );
	c->stream = c->own;
	return 0;
}

int rt_hip_upload_tree( rt_hip_ctx *c, const rt_hip_kdnode *nodes, uint64_t nNodes, const rt_hip_trirec *recs,
	uint64_t nRecs, const rt_hip_trishade *shade, uint64_t nTris, const rt_box *bbox, int depth )
{
	return rt_hip_upload_tree_lists( c, nodes, nNodes, recs, nRecs, 0, nullptr, shade, nTris, bbox, depth );
}

int rt_hip_upload_tree_lists( rt_hip_ctx *c, const rt_hip_kdnode *nodes, uint64_t nNodes, const rt_hip_trirec *recs,
	uint64_t nRecs, uint64_t nRecsFull, const rt_hip_leaf_full *leafFull, const rt_hip_trishade *shade, uint64_t nTris,
	const rt_box *bbox, int depth )
{
	if ( !c || !nodes || !bbox || nNodes == 0 || ( nRecs && !recs ) || ( nTris && !shade ) || ( nRecsFull && !leafFull ) )
		return fail( "rt_hip_upload_tree: bad arguments" );
	if ( !leafFull ) nRecsFull = 0;
	if ( depth > RT_HIP_MAX_TREE_DEPTH )
		return fail( "rt_hip_upload_tree: tree depth %d exceeds the traversal stack (%d)", depth, RT_HIP_MAX_TREE_DEPTH );
	const uint64_t nAll = nRecs + nRecsFull;
	if ( nNodes >= ( 1ull << 30 ) || nAll >= ( 1ull << 30 ) || nTris >= ( 1ull << 31 ) )
		return fail( "rt_hip_upload_tree: tree too large for 32-bit node links" );
	// host-side shape check: every link and record range must stay inside the
	// arrays (an out-of-range link would fault the GPU), and children must
	// come after their parent so no traversal can loop
	// The traversal stack is sized by the depth, so the depth is measured here
	// and not taken from the caller: level[i] = longest path from the root.
	std::vector<uint8_t> level( nNodes, 0 );
	for ( uint64_t i = 0; i < nNodes; ++i )
	{
		uint32_t b = nodes[i].b;
		if ( ( b & 3u ) == 3u )
		{
			if ( (uint64_t) nodes[i].a + ( b >> 2 ) > nRecs )
				return fail( "rt_hip_upload_tree: leaf %llu references records beyond %llu", (unsigned long long) i, (unsigned long long) nRecs );
			if ( leafFull && (uint64_t) leafFull[i].first + leafFull[i].count > nAll )
				return fail( "rt_hip_upload_tree: the complete list of leaf %llu lies beyond the %llu records", (unsigned long long) i, (unsigned long long) nAll );
			continue;
		}
		const uint64_t right = b >> 2;
		if ( right >= nNodes || right <= i || i + 1 >= nNodes )
			return fail( "rt_hip_upload_tree: node %llu has an invalid child link", (unsigned long long) i );
		if ( level[i] >= RT_HIP_MAX_TREE_DEPTH )
			return fail( "rt_hip_upload_tree: the tree is deeper than the traversal stack (%d)", RT_HIP_MAX_TREE_DEPTH );
		const uint8_t l = (uint8_t) ( level[i] + 1 );
		if ( level[i + 1] < l ) level[i + 1] = l;
		if ( level[right] < l ) level[right] = l;
	}
	uint64_t maxMat = 0;
	for ( uint64_t i = 0; i < nAll; ++i )
	{
		if ( recs[i].tri >= nTris )
			return fail( "rt_hip_upload_tree: record %llu references triangle %u of %llu", (unsigned long long) i, recs[i].tri, (unsigned long long) nTris );
		if ( recs[i].mat > maxMat ) maxMat = recs[i].mat;
	}
	for ( uint64_t i = 0; i < nTris; ++i )
		if ( shade[i].mat > maxMat ) maxMat = shade[i].mat;
	c->maxMatTree = maxMat;
	HIPCHK( hipSetDevice( c->device ) );
	scene_changed( c );
	if ( upload( c, c->nodes, nodes, nNodes * sizeof( rt_hip_kdnode ) ) ) return 1;
	// one spare record: the leaf loop loads one record ahead
	if ( ensure( c, c->recs, ( nAll + 1 ) * sizeof( rt_hip_trirec ) ) ) return 1;
	HIPCHK( hipMemsetAsync( (char *) c->recs.p + nAll * sizeof( rt_hip_trirec ), 0, sizeof( rt_hip_trirec ), c->stream ) );
	if ( upload( c, c->recs, recs, nAll * sizeof( rt_hip_trirec ) ) ) return 1;
	if ( upload( c, c->shade, shade, nTris * sizeof( rt_hip_trishade ) ) ) return 1;
	if ( leafFull && upload( c, c->leafFull, leafFull, nNodes * sizeof( rt_hip_leaf_full ) ) ) return 1;
	c->nNodes = nNodes; c->nRecs = nRecs; c->nTris = nTris; c->treeDepth = depth;
	c->S.nodes = (const uint2 *) c->nodes.p;
	c->S.recs = (const float4 *) c->recs.p;
	c->S.shade = (const float4 *) c->shade.p;
	c->S.leafFull = leafFull ? (const uint2 *) c->leafFull.p : nullptr;
	c->S.bbC[0] = bbox->center.x; c->S.bbC[1] = bbox->center.y; c->S.bbC[2] = bbox->center.z;
	c->S.bbE[0] = bbox->extents.x; c->S.bbE[1] = bbox->extents.y; c->S.bbE[2] = bbox->extents.z;
	return 0;
}

int rt_hip_adopt_tree( rt_hip_ctx *c, rt_kdtree_device *t )
{
	if ( !c || !t || !t->nodes || !t->recs || !t->shade || t->nNodes == 0 ) return fail( "rt_hip_adopt_tree: bad arguments" );
	if ( t->depth > RT_HIP_MAX_TREE_DEPTH )
		return fail( "rt_hip_adopt_tree: tree depth %d exceeds the traversal stack (%d)", t->depth, RT_HIP_MAX_TREE_DEPTH );
	if ( t->nNodes >= ( 1ull << 30 ) || t->nRecs >= ( 1ull << 30 ) || t->nTris >= ( 1ull << 31 ) )
		return fail( "rt_hip_adopt_tree: tree too large for 32-bit node links" );
	HIPCHK( hipSetDevice( c->device ) );
	// frames that still read the old tree must be done before it is freed (hipFree also waits for the
	// device, frames of borrowers included; borrowers must share again before their next frame)
	HIPCHK( hipStreamSynchronize( c->stream ) );
	scene_changed( c );
	DevBuf *dst[3] = { &c->nodes, &c->recs, &c->shade };
	void *src[3] = { t->nodes, t->recs, t->shade };
	const size_t bytes[3] = { t->nNodes * sizeof( rt_hip_kdnode ), ( t->nRecs + ( t->leafFull ? t->nRecsFull : 0 ) + 1 ) * sizeof( rt_hip_trirec ),
		( t->nTris ? t->nTris : 1 ) * sizeof( rt_hip_trishade ) };
	// The old tree: to the builder for its next result when nobody borrows this context's scene (a frame loop that
	// rebuilds every frame would otherwise free 0.4 GB per frame), else freed - hipFree waits for the whole device.
	void *old[4] = { c->nodes.p, c->recs.p, c->shade.p, c->leafFull.p };
	if ( c->tag && c->tag->refs.load() == 1 )
		rt_kdtree_device_recycle( old[0], old[1], old[2], old[3] );
	else
		for ( void *p : old )
			if ( p ) HIPCHK( hipFree( p ) );
	for ( int i = 0; i < 3; ++i )
	{
		dst[i]->p = src[i];
		dst[i]->cap = bytes[i];
	}
	c->leafFull.p = t->leafFull;
	c->leafFull.cap = t->leafFull ? t->nNodes * sizeof( rt_hip_leaf_full ) : 0;
	c->S.leafFull = (const uint2 *) c->leafFull.p;
	c->maxMatTree = t->maxMaterial;
	c->nNodes = t->nNodes; c->nRecs = t->nRecs; c->nTris = t->nTris; c->treeDepth = t->depth;
	c->S.nodes = (const uint2 *) c->nodes.p;
	c->S.recs = (const float4 *) c->recs.p;
	c->S.shade = (const float4 *) c->shade.p;
	c->S.bbC[0] = t->bbox.center.x; c->S.bbC[1] = t->bbox.center.y; c->S.bbC[2] = t->bbox.center.z;
	c->S.bbE[0] = t->bbox.extents.x; c->S.bbE[1] = t->bbox.extents.y; c->S.bbE[2] = t->bbox.extents.z;
	memset( t, 0, sizeof( *t ) );
	return 0;
}

int rt_hip_build_tree( rt_hip_ctx *c, const rt_vertex *verts, const rt_triangle *tris, uint64_t vertexCount,
	uint64_t trianglesCount, int maxDepth, int maxPrims )
{
	if ( !c ) return fail( "null context" );
	HIPCHK( hipSetDevice( c->device ) );
	rt_kdtree_device t;
	rt_kdtree_build_opts o;
	memset( &o, 0, sizeof( o ) );
	o.maxDepth = maxDepth;
	o.maxPrims = maxPrims;
	o.pruneLeaves = c->pruneLeaves ? RT_PRUNE_LEAVES_MARGIN : 0.0f;
	o.stageTicket = c->stageTicket;
	c->stageTicket = 0;
	int rc = rt_kdtree_build_device_ex( verts, tris, vertexCount, trianglesCount, &o, c->stream, &t );
	if ( rc )
		return fail( "rt_hip_build_tree: builder returned %d (%s)", rc, rt_kdtree_device_last_error() );
	if ( rt_hip_adopt_tree( c, &t ) )
	{
		rt_kdtree_device_release( &t );
		return 1;
	}
	return 0;
}

int rt_hip_stage_geometry( rt_hip_ctx *c, const rt_vertex *verts, uint64_t nVerts, const rt_triangle *tris, uint64_t nTris )
{
	if ( !c ) return fail( "null context" );
	HIPCHK( hipSetDevice( c->device ) );
	const bool begun = c->stageTicket != 0;
	if ( int rc = rt_kdtree_device_stage( &c->stageTicket, verts, nVerts, tris, nTris, c->stream ) )
	{
		c->stageTicket = 0;
		return fail( "rt_hip_stage_geometry: builder returned %d (%s)", rc, rt_kdtree_device_last_error() );
	}
	// (the builder's buffers went to other geometry since the last call: nothing is staged any more)
	return begun && c->stageTicket == 0 ? 2 : 0;
}

void rt_hip_unstage_geometry( rt_hip_ctx *c )
{
	if ( c )
		c->stageTicket = 0;
}

// A second context on the same GPU that renders the same scene (frames in
// flight: frame i + 1 starts on its own streams and work buffers while frame i
// drains) reads the tree, materials, spheres and lights of `from` instead of
// holding copies: one working set in the caches, not two.
int rt_hip_share_scene( rt_hip_ctx *c, const rt_hip_ctx *from )
{
	if ( !c || !from || c == from ) return fail( "rt_hip_share_scene: bad arguments" );
	if ( c->device != from->device ) return fail( "rt_hip_share_scene: contexts on different devices" );
	if ( !from->S.nodes || !from->S.mats || !from->S.spheres || !from->S.lights ) return fail( "rt_hip_share_scene: the source context holds no complete scene" );
	HIPCHK( hipSetDevice( c->device ) );
	HIPCHK( hipStreamSynchronize( c->stream ) );
	// own copies, if any, are released: the pointers below replace them
	DevBuf *own[] = { &c->nodes, &c->recs, &c->shade, &c->mats, &c->spheres, &c->sphereMat, &c->lights, &c->leafFull };
	for ( DevBuf *b : own )
		if ( b->p ) { HIPCHK( hipFree( b->p ) ); b->p = nullptr; b->cap = 0; }
	c->S.nodes = from->S.nodes; c->S.recs = from->S.recs; c->S.shade = from->S.shade; c->S.leafFull = from->S.leafFull;
	c->S.mats = from->S.mats; c->S.spheres = from->S.spheres; c->S.sphereMat = from->S.sphereMat; c->S.lights = from->S.lights;
	c->S.nSpheres = from->S.nSpheres; c->S.nLights = from->S.nLights; c->S.alphaMonotone = from->S.alphaMonotone;
	memcpy( c->S.bbC, from->S.bbC, sizeof( c->S.bbC ) );
	memcpy( c->S.bbE, from->S.bbE, sizeof( c->S.bbE ) );
	c->nNodes = from->nNodes; c->nRecs = from->nRecs; c->nTris = from->nTris; c->nMats = from->nMats;
	c->maxMatTree = from->maxMatTree; c->maxMatSphere = from->maxMatSphere; c->treeDepth = from->treeDepth;
	scene_tag_release( c->lender );
	c->lender = from->tag;
	c->lender->refs.fetch_add( 1 );
	c->lenderGen = c->lender->gen.load();
	scene_changed( c );                              // (whoever borrowed from c)
	return 0;
}

int rt_hip_upload_materials( rt_hip_ctx *c, const rt_material *m, uint64_t n )
{
	if ( !c || ( n && !m ) ) return fail( "rt_hip_upload_materials: bad arguments" );
	HIPCHK( hipSetDevice( c->device ) );
	scene_changed( c );
	if ( upload( c, c->mats, m, n * sizeof( rt_material ) ) ) return 1;
	c->nMats = n;
	c->S.alphaMonotone = 1;
	for ( uint64_t i = 0; i < n; ++i )
		if ( !( m[i].color.a >= 0.0f ) )
			c->S.alphaMonotone = 0;
	c->S.mats = (const rt_material *) c->mats.p;
	return 0;
}

int rt_hip_upload_spheres( rt_hip_ctx *c, const rt_sphere *s, uint64_t n )
{
	if ( !c || ( n && !s ) ) return fail( "rt_hip_upload_spheres: bad arguments" );
	HIPCHK( hipSetDevice( c->device ) );
	std::vector<float> g( 4 * n + 4 );
	std::vector<uint32_t> mi( n + 1 );
	c->maxMatSphere = 0;
	for ( uint64_t i = 0; i < n; ++i )
	{
		if ( s[i].mat >= ( 1ull << 31 ) ) return fail( "rt_hip_upload_spheres: material slot out of range" );
		if ( s[i].mat > c->maxMatSphere ) c->maxMatSphere = s[i].mat;
		g[4 * i] = s[i].pos.x; g[4 * i + 1] = s[i].pos.y; g[4 * i + 2] = s[i].pos.z; g[4 * i + 3] = s[i].rad;
		mi[i] = (uint32_t) s[i].mat;
	}
	scene_changed( c );
	if ( upload( c, c->spheres, g.data(), 16 * n ) ) return 1;
	if ( upload( c, c->sphereMat, mi.data(), 4 * n ) ) return 1;
	c->S.spheres = (const float4 *) c->spheres.p;
	c->S.sphereMat = (const uint32_t *) c->sphereMat.p;
	c->S.nSpheres = (int) n;
	return 0;
}

int rt_hip_upload_lights( rt_hip_ctx *c, const rt_point_light *l, uint64_t n )
{
	if ( !c || ( n && !l ) ) return fail( "rt_hip_upload_lights: bad arguments" );
	HIPCHK( hipSetDevice( c->device ) );
	std::vector<float> g( 12 * n + 12, 0.0f );
	for ( uint64_t i = 0; i < n; ++i )
	{
		float *p = g.data() + 12 * i;
		p[0] = l[i].col.b; p[1] = l[i].col.g; p[2] = l[i].col.r; p[3] = l[i].col.a;
		p[4] = l[i].pos.x; p[5] = l[i].pos.y; p[6] = l[i].pos.z;
		p[8] = l[i].rad;
	}
	scene_changed( c );
	if ( upload( c, c->lights, g.data(), 48 * n ) ) return 1;
	c->S.lights = (const float4 *) c->lights.p;
	c->S.nLights = (int) n;
	return 0;
}

int rt_hip_set_camera( rt_hip_ctx *c, const rt_camera *cam )
{
	if ( !c || !cam ) return fail( "rt_hip_set_camera: bad arguments" );
	memcpy( c->S.world, &cam->world, 64 );
	memcpy( c->S.vtp, &cam->viewToPersp, 64 );
	return 0;
}

static void update_local_rows( rt_hip_ctx *c )
{
	DevScene &S = c->S;
	int rows = 0;
	if ( S.h > 0 )
	{
		int bands = ( S.h + S.bandH - 1 ) / S.bandH;
		for ( int b = S.part; b < bands; b += S.nParts )
		{
			int r = S.h - b * S.bandH;
			rows += r < S.bandH ? r : S.bandH;
		}
	}
	// compact storage keeps full bands except possibly the last one; a short
	// last band can only be the final local band, so rows == local extent
	S.localH = rows;
	c->workDirty = true;
}

int rt_hip_set_image( rt_hip_ctx *c, int w, int h )
{
	if ( !c || w < 2 || h < 2 || (long long) w * h >= ( 1ll << 30 ) ) return fail( "rt_hip_set_image: bad size %dx%d", w, h );
	c->S.w = w;
	c->S.h = h;
	update_local_rows( c );
	return 0;
}

int rt_hip_set_tile( rt_hip_ctx *c, int bandH, int nParts, int part )
{
	if ( !c || bandH < 1 || nParts < 1 || part < 0 || part >= nParts ) return fail( "rt_hip_set_tile: bad arguments" );
	c->S.bandH = bandH;
	c->S.nParts = nParts;
	c->S.part = part;
	update_local_rows( c );
	return 0;
}

int rt_hip_local_rows( rt_hip_ctx *c ) { return c ? c->S.localH : 0; }

int rt_hip_set_param( rt_hip_ctx *c, int key, int value )
{
	if ( !c ) return fail( "null context" );
	if ( key == RT_PARAM_STACK_SIZE )
	{
		if ( value < 1 || value > RT_STACK_SIZE_MAX ) return fail( "RT_PARAM_STACK_SIZE must be 1..%d", RT_STACK_SIZE_MAX );
		c->S.stackSize = value;
		c->workDirty = true;
		return 0;
	}
	if ( key == RT_PARAM_OVERLAP )
	{
		c->overlap = value ? 1 : 0;
		return 0;
	}
	if ( key == RT_PARAM_COOP_LIVE )
	{
		if ( value < 0 || value > 32 ) return fail( "RT_PARAM_COOP_LIVE must be 0..32" );
		c->coopLive = value;
		return 0;
	}
	if ( key == RT_PARAM_LAUNCH_WAVES )
	{
		if ( value < 1 || value > RT_MAX_LAUNCH_WAVES ) return fail( "RT_PARAM_LAUNCH_WAVES must be 1..%d", RT_MAX_LAUNCH_WAVES );
		c->launchWaves = value;
		return 0;
	}
	if ( key == RT_PARAM_PRUNE_LEAVES )
	{
		c->pruneLeaves = value ? 1 : 0;
		return 0;
	}
	if ( key == RT_PARAM_COOP_STACK )
	{
		if ( value < 0 || value > RT_LDS_STACK ) return fail( "RT_PARAM_COOP_STACK must be 0..%d", RT_LDS_STACK );
		c->coopCap = value;
		return 0;
	}
#ifdef RT_EXP_REPLAY
	if ( key == 100 )
	{
		c->replay = value;
		if ( value && !c->evRepStart )
		{
			HIPCHK( hipSetDevice( c->device ) );
			HIPCHK( hipEventCreateWithFlags( &c->evRepStart, hipEventDisableTiming ) );
			for ( int k = 0; k < RT_MAX_SLOTS; ++k )
				HIPCHK( hipEventCreateWithFlags( &c->evRepDone[k], hipEventDisableTiming ) );
		}
		return 0;
	}
#endif
	return fail( "rt_hip_set_param: unknown key %d", key );
}

// Entries per ray / hit / hand-over list.  A list holds at most npix valid
// entries.  Long lists are appended through wave-private reservations of
// RT_RESV_BLOCK entries: a reservation is given up with fewer than 64 of its
// entries unused (hence the 4/3), and every wave of every launch that appends to
// the list can leave one unfinished reservation behind.  A ray list is fed by
// the closest-hit passes of up to four parent slots.
// Reservations are only used by launches whose input has at least 256 entries
// per wave, so small frames need no room for them.  Appends check the capacity
// on the device all the same (Counters::overflow).
static size_t list_capacity( const rt_hip_ctx *c, size_t npix )
{
	static_assert( RT_RESV_BLOCK >= 256u || RT_RESV_BLOCK == 0u, "the 4/3 bound below assumes blocks of at least 256 entries" );
	size_t cap = npix + npix / 3 + 64;
	const size_t wavesPerLaunch = (size_t) c->maxGrid * ( RT_BLOCK / 64 );
	if ( npix >= 256 )
	{
		// launches with reservations need count >= waves * 256; grids never exceed maxGrid
		size_t waves = cap / 256 + 1;
		if ( waves > wavesPerLaunch ) waves = wavesPerLaunch;
		int feeders = c->S.stackSize - 1;
		if ( feeders > 4 ) feeders = 4;
		if ( feeders < 1 ) feeders = 1;
		cap += (size_t) feeders * waves * RT_RESV_BLOCK;
	}
	return cap;
}

static int prepare_work( rt_hip_ctx *c )
{
	DevScene &S = c->S;
	Work &W = c->W;
	if ( S.w < 2 || S.h < 2 ) return fail( "image size not set" );
	size_t npix = (size_t) S.w * (size_t) S.localH;
	if ( npix == 0 ) npix = 1;
	int slots = S.stackSize;
	if ( ensure( c, c->ctr, sizeof( Counters ) ) ) return 1;
	// A list holds at most npix valid entries.  A reservation is given up with
	// fewer than 64 of its RT_RESV_BLOCK entries unused, and every wave leaves at
	// most one unfinished reservation behind: that bounds the holes.
	const size_t cap = list_capacity( c, npix );
#ifdef RT_EXP_REPLAY
	if ( ensure( c, c->rays, 2 * 4 * sizeof( float4 ) * cap * (size_t) ( slots > 1 ? slots - 1 : 1 ) ) ) return 1;
#else
	if ( ensure( c, c->rays, 4 * sizeof( float4 ) * cap * (size_t) ( slots > 1 ? slots - 1 : 1 ) ) ) return 1;
#endif
	if ( ensure( c, c->hitA, sizeof( float4 ) * cap * slots ) ) return 1;
	if ( ensure( c, c->hitB, sizeof( uint2 ) * cap * slots ) ) return 1;
	if ( ensure( c, c->hitN, sizeof( float4 ) * cap * slots ) ) return 1;
	if ( ensure( c, c->res, sizeof( float4 ) * npix * slots ) ) return 1;
	if ( ensure( c, c->fac, sizeof( float4 ) * npix * slots ) ) return 1;
	if ( ensure( c, c->shadow, sizeof( float ) * cap * slots * (size_t) ( S.nLights > 0 ? S.nLights : 1 ) ) ) return 1;
	if ( ensure( c, c->stackPos, sizeof( uint32_t ) * npix ) ) return 1;
	if ( ensure( c, c->hitId, sizeof( int32_t ) * npix ) ) return 1;
	if ( ensure( c, c->frame, sizeof( rt_argb ) * npix ) ) return 1;
	// one strip for the closest-hit chain and one per auxiliary stream
#ifdef RT_EXP_REPLAY
	if ( ensure( c, c->spill, ( 1 + RT_AUX_STREAMS + RT_MAX_SLOTS ) * sizeof( uint2 ) * (size_t) c->maxGrid * RT_BLOCK * RT_SPILL_CAP ) ) return 1;
#else
	if ( ensure( c, c->spill, ( 1 + RT_AUX_STREAMS ) * sizeof( uint2 ) * (size_t) c->maxGrid * RT_BLOCK * RT_SPILL_CAP ) ) return 1;
#endif
	size_t per = cap * (size_t) ( slots > 1 ? slots - 1 : 1 );
	W.ctr = (Counters *) c->ctr.p;
	W.rayA = (float4 *) c->rays.p;
	W.rayB = W.rayA + per;
	W.rayC = W.rayB + per;
	W.rayD = W.rayC + per;
#ifdef RT_EXP_REPLAY
	W.altOffset = 4 * per;
	W.replay = 0u;
#endif
	W.hitA = (float4 *) c->hitA.p;
	W.hitB = (uint2 *) c->hitB.p;
	W.hitN = (float4 *) c->hitN.p;
	W.res = (float4 *) c->res.p;
	W.fac = (float4 *) c->fac.p;
	W.shadow = (float *) c->shadow.p;
	c->workLights = S.nLights;
	W.stackPos = (uint32_t *) c->stackPos.p;
	W.hitId = (int32_t *) c->hitId.p;
	W.spill = (uint2 *) c->spill.p;
	W.npix = (uint32_t) ( (size_t) S.w * (size_t) S.localH );
	W.cap = (uint32_t) cap;
	W.tilesX = (uint32_t) ( ( S.w + 7 ) / 8 );
	W.nChunks0 = W.tilesX * (uint32_t) ( ( S.localH + 7 ) / 8 );
	W.spillStride = (uint32_t) c->maxGrid * RT_BLOCK;
	c->validPixels = W.npix;
	c->workDirty = false;
	return 0;
}

// the per-frame constants of DevScene that follow from its other fields (same fp32 operations, in the same
// order, as the expressions they replace in the kernels: this file is compiled without contraction)
static void derive_scene( DevScene &S )
{
	for ( int k = 0; k < 3; ++k )
	{
		S.bbLo[k] = S.bbC[k] - S.bbE[k];
		S.bbHi[k] = S.bbC[k] + S.bbE[k];
	}
	S.xTr = 2.0f / (float) ( S.w - 1 );
	S.yTr = 2.0f / (float) ( S.h - 1 );
	const float *W = S.world;
	const float sx = 0.0f, sy = 0.0f, sz = 0.0f;
	S.camSrc[0] = W[0] * sx + W[4] * sy + W[8] * sz + W[12];
	S.camSrc[1] = W[1] * sx + W[5] * sy + W[9] * sz + W[13];
	S.camSrc[2] = W[2] * sx + W[6] * sy + W[10] * sz + W[14];
}

static int enqueue_frame( rt_hip_ctx *c, const Work &W, rt_argb *out, unsigned flags, bool overlap, int gridCap );
#ifdef RT_EXP_REPLAY
static int enqueue_replay( rt_hip_ctx *c, const Work &W, rt_argb *out );
#endif

int rt_hip_render( rt_hip_ctx *c, void *d_out, unsigned flags )
{
	if ( !c ) return fail( "null context" );
	if ( lender_unchanged( c, "rt_hip_render" ) ) return 1;
	if ( !c->S.nodes ) return fail( "rt_hip_render: no tree uploaded" );
	if ( !c->S.mats ) return fail( "rt_hip_render: no materials uploaded" );
	// a material slot beyond the uploaded table would be an out-of-bounds read on the GPU
	if ( ( c->nRecs && c->maxMatTree >= c->nMats ) || ( c->S.nSpheres && c->maxMatSphere >= c->nMats ) )
		return fail( "rt_hip_render: geometry references material slot %llu but only %llu are uploaded",
			(unsigned long long) ( c->maxMatTree > c->maxMatSphere ? c->maxMatTree : c->maxMatSphere ), (unsigned long long) c->nMats );
	HIPCHK( hipSetDevice( c->device ) );
	if ( c->S.nSpheres == 0 && !c->S.spheres ) { rt_sphere z; memset( &z, 0, sizeof( z ) ); if ( rt_hip_upload_spheres( c, &z, 0 ) ) return 1; }
	if ( c->S.nLights == 0 && !c->S.lights ) { rt_point_light z; memset( &z, 0, sizeof( z ) ); if ( rt_hip_upload_lights( c, &z, 0 ) ) return 1; }
	if ( (unsigned long long) list_capacity( c, (size_t) c->S.w * c->S.localH ) * (unsigned long long) ( c->S.nLights > 0 ? c->S.nLights : 1 ) >= ( 1ull << 31 ) )
		return fail( "rt_hip_render: pixels x lights exceeds the 31-bit work-item range" );
	if ( ( c->workDirty || c->workLights != c->S.nLights ) && prepare_work( c ) ) return 1;
#ifdef RT_EXP_REPLAY
	if ( c->replay ) return enqueue_replay( c, c->W, d_out ? (rt_argb *) d_out : (rt_argb *) c->frame.p );
#endif
	return enqueue_frame( c, c->W, d_out ? (rt_argb *) d_out : (rt_argb *) c->frame.p, flags, c->overlap && c->S.stackSize > 1, 0 );
}

// The launches of one frame over the work buffers `W` (see the top of this
// file).  gridCap > 0 limits every persistent grid to that many workgroups.
// Allocates nothing unless RT_HIP_TIMING asks for events that do not exist yet.
static int enqueue_frame( rt_hip_ctx *c, const Work &W, rt_argb *out, unsigned flags, bool overlap, int gridCap )
{
	const bool stats = flags & RT_HIP_STATS, ids = flags & RT_HIP_HITIDS, timing = flags & RT_HIP_TIMING;
	derive_scene( c->S );
	const DevScene &S = c->S;
	hipStream_t s = c->stream;

	if ( timing && !c->haveEvents )
	{
		for ( auto &f : c->ev ) for ( auto &e : f ) HIPCHK( hipEventCreate( &e ) );
		c->haveEvents = true;
	}
	if ( timing && c->timedSlots != S.stackSize )
	{
		c->timedSlots = S.stackSize;
		c->timedFrames = 0;
	}
	hipEvent_t *ev = c->ev[c->timedFrames % RT_TIMED_FRAMES];
	hipEvent_t *evFrame = ev + 6 * RT_MAX_SLOTS + 2;
	const bool big = W.npix >= c->bigFrame;
	const int lights = S.nLights > 0 ? S.nLights : 1;

	// Timing records one event behind every kernel and none in front of it: a kernel starts when the
	// last of its predecessors has ended (the kernel before it on its stream, and the event its stream
	// waited for), which the read-out works out from the end events.  In the overlapped schedule the
	// end events of k_closest<k> and k_light<k> are the very events the fork and the join need, so a
	// timed frame puts 8 markers more into the streams than an untimed one, not 34.
	if ( timing ) HIPCHK( hipEventRecord( evFrame[0], s ) );
	c->lastCtr = W.ctr;
	c->lastCap = W.cap;
	hipLaunchKernelGGL( k_put_camera, dim3( 1 ), dim3( 64 ), 0, s, S, (float *) c->cam.p );
	HIPCHK( hipMemsetAsync( W.ctr, 0, sizeof( Counters ), s ) );
	if ( W.npix == 0 )
		return 0;
	if ( timing )
	{
		c->evOverlap[c->timedFrames % RT_TIMED_FRAMES] = overlap;
		HIPCHK( hipEventRecord( ev[0], s ) );          // start of k_closest<0> (behind the memset)
	}
	for ( int k = 0; k < S.stackSize; ++k )
	{
		hipEvent_t *e = ev + 6 * k;
		hipStream_t x = overlap ? c->aux[k % RT_AUX_STREAMS] : s;
		// every traversal running at the same time needs its own stack-overflow strip
		Work Wx = W;
		if ( overlap ) Wx.spill = W.spill + (size_t) ( 1 + k % RT_AUX_STREAMS ) * W.spillStride * RT_SPILL_CAP;
		Wx.coopLive = c->coopShadow && c->coopLive ? (uint32_t) c->coopLive | ( (uint32_t) c->coopFill << 8 ) | ( c->coopAtOnce ? 0x80000000u : 0u ) : 0u;
		Wx.coopCap = (uint32_t) c->coopCap;

		// a persistent grid never needs more waves than there are chunks
		const uint32_t wpb = RT_BLOCK / 64u, want = k == 0 ? ( W.nChunks0 + wpb - 1u ) / wpb : 0u;
		int gc = c->gridClosest[big][stats], gl = c->gridLight;
		// beside the chain a shadow pass leaves part of every CU to it
		int gw = overlap && c->occShadow[stats] > RT_AUX_OCC ? RT_AUX_OCC * c->numCU : c->gridShadow[stats];
		const int most = c->launchWaves * c->numCU / ( RT_BLOCK / 64 );
		if ( gc > most ) gc = most;
		if ( gw > most ) gw = most;
		if ( k == 0 && want < (uint32_t) gc ) gc = (int) ( want ? want : 1u );
		if ( k == 0 && want * (uint32_t) lights < (uint32_t) gw ) gw = (int) ( want ? want * lights : 1u );
		if ( k == 0 && want < (uint32_t) gl ) gl = (int) ( want ? want : 1u );
		if ( gridCap > 0 )
		{
			if ( gc > gridCap ) gc = gridCap;
			if ( gw > gridCap ) gw = gridCap;
			if ( gl > gridCap ) gl = gridCap;
		}

		void ( *closest )( DevScene, Work, int, int ) = big
			? ( stats ? k_closest<true, RT_WAVES_CLOSEST_STATS> : k_closest<false, RT_WAVES_CLOSEST_BIG> )
			: ( stats ? k_closest<true, RT_WAVES_CLOSEST_STATS> : k_closest<false, RT_WAVES_CLOSEST> );
		Work Wc = W;
		Wc.coopLive = c->coopClosest && c->coopLive ? (uint32_t) c->coopLive | ( (uint32_t) c->coopFill << 8 ) | ( c->coopAtOnce ? 0x80000000u : 0u ) : 0u;
		Wc.coopCap = (uint32_t) c->coopCap;
		hipLaunchKernelGGL( closest, dim3( gc ), dim3( RT_BLOCK ), 0, s, S, Wc, k, ids ? 1 : 0 );
		hipEvent_t hit = timing ? e[1] : c->evHit[k];
		if ( timing || overlap ) HIPCHK( hipEventRecord( hit, s ) );
		if ( overlap ) HIPCHK( hipStreamWaitEvent( x, hit, 0 ) );

		if ( stats ) hipLaunchKernelGGL( k_shadow<true>, dim3( gw ), dim3( RT_BLOCK ), 0, x, S, Wx, k );
		else hipLaunchKernelGGL( k_shadow<false>, dim3( gw ), dim3( RT_BLOCK ), 0, x, S, Wx, k );
		if ( timing ) HIPCHK( hipEventRecord( e[3], x ) );

		hipLaunchKernelGGL( k_light, dim3( gl ), dim3( RT_BLOCK ), 0, x, S, W, k );
		hipEvent_t lit = timing ? e[5] : c->evLit[k];
		if ( timing || overlap ) HIPCHK( hipEventRecord( lit, x ) );
	}
	if ( overlap )
		for ( int k = 0; k < S.stackSize; ++k )
			HIPCHK( hipStreamWaitEvent( s, timing ? ev[6 * k + 5] : c->evLit[k], 0 ) );
	hipLaunchKernelGGL( k_fold, dim3( ( W.npix + RT_BLOCK - 1 ) / RT_BLOCK ), dim3( RT_BLOCK ), 0, s, W, out, S.stackSize );
	if ( timing ) HIPCHK( hipEventRecord( evFrame[1], s ) );
	if ( timing ) ++c->timedFrames;
	HIPCHK( hipGetLastError() );
	return 0;
}

#ifdef RT_EXP_REPLAY
// Timing experiment (development build): what would a frame cost if no pass had to wait for the pass of the slot
// before it?  The ray lists of the frame rendered last on this context are still in place; every slot's closest-hit
// pass is run again over its list, the slots side by side on the context's three streams (more streams per context
// make the runtime time-slice its hardware queues), each followed by its shadow and lighting pass; child rays are
// appended to throw-away lists.  The passes do exactly the work of a normal frame
// (same lists, same hits, same image); only the order constraints between slots are gone - including the true ones
// (a child ray after its parent), so this bounds from below what any schedule of the same passes can reach.
static int enqueue_replay( rt_hip_ctx *c, const Work &W0, rt_argb *out )
{
	derive_scene( c->S );
	const DevScene &S = c->S;
	hipStream_t s = c->stream;
	Work W = W0;
	W.replay = 1u;
	// everything but the lengths of the ray lists starts again
	Counters *d = W.ctr;
	HIPCHK( hipMemsetAsync( (char *) d + offsetof( Counters, headClosest ), 0, offsetof( Counters, overflow ) - offsetof( Counters, headClosest ), s ) );
	HIPCHK( hipMemsetAsync( d->rayCountAlt, 0, sizeof( d->rayCountAlt ), s ) );
	hipLaunchKernelGGL( k_put_camera, dim3( 1 ), dim3( 64 ), 0, s, S, (float *) c->cam.p );
	HIPCHK( hipEventRecord( c->evRepStart, s ) );
	const int lights = S.nLights > 0 ? S.nLights : 1;
	for ( int k = 0; k < S.stackSize; ++k )
	{
		const int lane = k % ( 1 + RT_AUX_STREAMS );
		hipStream_t x = lane == 0 ? s : c->aux[lane - 1];
		if ( lane != 0 && k < 1 + RT_AUX_STREAMS ) HIPCHK( hipStreamWaitEvent( x, c->evRepStart, 0 ) );
		Work Wk = W;
		Wk.spill = W.spill + (size_t) ( 1 + RT_AUX_STREAMS + k ) * W.spillStride * RT_SPILL_CAP;
		Wk.coopLive = c->coopLive ? (uint32_t) c->coopLive | ( (uint32_t) c->coopFill << 8 ) | ( c->coopAtOnce ? 0x80000000u : 0u ) : 0u;
		Wk.coopCap = (uint32_t) c->coopCap;
		const uint32_t want = k == 0 ? W.nChunks0 : 0u;
		int gc = c->gridClosest[0][0], gw = c->gridShadow[0], gl = c->gridLight;
		const int most = c->launchWaves * c->numCU;
		if ( gc > most ) gc = most;
		if ( gw > most ) gw = most;
		if ( k == 0 && want < (uint32_t) gc ) gc = (int) ( want ? want : 1u );
		if ( k == 0 && want * (uint32_t) lights < (uint32_t) gw ) gw = (int) ( want ? want * lights : 1u );
		if ( k == 0 && want < (uint32_t) gl ) gl = (int) ( want ? want : 1u );
		hipLaunchKernelGGL( ( k_closest<false, RT_WAVES_CLOSEST> ), dim3( gc ), dim3( RT_BLOCK ), 0, x, S, Wk, k, 0 );
		hipLaunchKernelGGL( k_shadow<false>, dim3( gw ), dim3( RT_BLOCK ), 0, x, S, Wk, k );
		hipLaunchKernelGGL( k_light, dim3( gl ), dim3( RT_BLOCK ), 0, x, S, Wk, k );
		if ( lane != 0 ) HIPCHK( hipEventRecord( c->evRepDone[k], x ) );
	}
	for ( int k = 0; k < S.stackSize; ++k )
		if ( k % ( 1 + RT_AUX_STREAMS ) != 0 ) HIPCHK( hipStreamWaitEvent( s, c->evRepDone[k], 0 ) );
	hipLaunchKernelGGL( k_fold, dim3( ( W.npix + RT_BLOCK - 1 ) / RT_BLOCK ), dim3( RT_BLOCK ), 0, s, W, out, S.stackSize );
	HIPCHK( hipGetLastError() );
	return 0;
}
#endif

static int check_overflow( rt_hip_ctx *c );

// Waits for everything enqueued on the context's stream and reports a frame whose work lists ran out of
// room (either entry point; the flag lives in the counters of the frame enqueued last).
int rt_hip_sync( rt_hip_ctx *c )
{
	if ( !c ) return fail( "null context" );
	HIPCHK( hipSetDevice( c->device ) );
	HIPCHK( hipStreamSynchronize( c->stream ) );
	return check_overflow( c );
}

// ---------------------------------------------------------------------------
// rt_hip_render_noalloc: a frame without a single allocation
//
// Callers that cannot allocate (HIP graph capture, tight serving loops,
// harnesses that forbid hipMalloc) get the same launches over work buffers
// carved from one file-scope device array.  Every GPU the library runs on holds
// its own copy of the array, at its own address, until the process ends; the
// address is looked up once per device (rt_hip_create) and only that pointer is
// ever handed to kernels or memsets.  All calls on one GPU share the array:
// frames that overlap in time on one GPU race, so the entry point is
// single-flight per GPU: a call on another stream than the previous call's is
// refused while that previous frame has not completed (noalloc_flight).  The
// counters are cleared at the start of every frame; nothing else needs to start
// at zero (every list is bounded by its own counter, every per-pixel record is
// written before it is read).
// ---------------------------------------------------------------------------
#define RT_NOALLOC_GRID ( 32768 / RT_BLOCK )      // most workgroups per persistent launch: 512 waves (sizes the stack-overflow strip)

constexpr size_t noalloc_align( size_t x ) { return ( x + 255 ) & ~(size_t) 255; }
struct NoallocLayout
{
	// entries per list: the frames of this entry point append without block
	// reservations (Work::exactAppends - whatever the grid a launch ends up with,
	// and whatever the padded length of the slot-0 list of a narrow image), so a
	// list has no holes and holds at most one entry per pixel
	static constexpr size_t cap = RT_HIP_NOALLOC_MAX_PIXELS + 64;
	static constexpr size_t npix = RT_HIP_NOALLOC_MAX_PIXELS;
	static constexpr size_t slots = RT_HIP_NOALLOC_MAX_SLOTS, lights = RT_HIP_NOALLOC_MAX_LIGHTS;
	static constexpr size_t oCtr = 0;
	static constexpr size_t oRays = oCtr + noalloc_align( sizeof( Counters ) );
	static constexpr size_t oHitA = oRays + noalloc_align( 4 * sizeof( float4 ) * cap * ( slots - 1 ) );
	static constexpr size_t oHitB = oHitA + noalloc_align( sizeof( float4 ) * cap * slots );
	static constexpr size_t oHitN = oHitB + noalloc_align( sizeof( uint2 ) * cap * slots );
	static constexpr size_t oRes = oHitN + noalloc_align( sizeof( float4 ) * cap * slots );
	static constexpr size_t oFac = oRes + noalloc_align( sizeof( float4 ) * npix * slots );
	static constexpr size_t oShadow = oFac + noalloc_align( sizeof( float4 ) * npix * slots );
	static constexpr size_t oStackPos = oShadow + noalloc_align( sizeof( float ) * cap * slots * lights );
	static constexpr size_t oHitId = oStackPos + noalloc_align( sizeof( uint32_t ) * npix );
	static constexpr size_t oSpill = oHitId + noalloc_align( sizeof( int32_t ) * npix );
	static constexpr size_t total = oSpill + noalloc_align( sizeof( uint2 ) * (size_t) RT_NOALLOC_GRID * RT_BLOCK * RT_SPILL_CAP );
};
static_assert( RT_HIP_NOALLOC_MAX_SLOTS >= 2 && RT_HIP_NOALLOC_MAX_SLOTS <= RT_MAX_SLOTS, "slot limit of the allocation-free entry point" );
static_assert( NoallocLayout::total <= RT_HIP_NOALLOC_BYTES, "RT_HIP_NOALLOC_BYTES in rt_hip.h must cover the layout" );

static __device__ __align__( 256 ) unsigned char scratch[RT_HIP_NOALLOC_BYTES];

// device address of `scratch` on each GPU (the array's name is only a handle on the host)
static unsigned char *g_scratch_ptr[RT_MAX_DEVICES] = {};
// the allocation-free frame enqueued last on each GPU: its stream, and an event behind it
// (created with the scratch look-up, i.e. at set-up, not by the entry point)
struct NoallocFlight { hipEvent_t done = nullptr; hipStream_t stream = nullptr; bool pending = false; };
static NoallocFlight g_flight[RT_MAX_DEVICES];

static int scratch_lookup( int device )
{
	if ( device < 0 || device >= RT_MAX_DEVICES ) return fail( "device ordinal %d beyond the %d this library keeps scratch addresses for", device, RT_MAX_DEVICES );
	if ( g_scratch_ptr[device] )
		return 0;
	unsigned char *p = nullptr;
	HIPCHK( hipSetDevice( device ) );
	HIPCHK( hipGetSymbolAddress( (void **) &p, HIP_SYMBOL( scratch ) ) );
	if ( !p ) return fail( "hipGetSymbolAddress returned a null address for the scratch array" );
	HIPCHK( hipEventCreateWithFlags( &g_flight[device].done, hipEventDisableTiming ) );
	g_scratch_ptr[device] = p;
	return 0;
}

// Single flight per GPU: every allocation-free frame on a GPU uses that GPU's one work area.  Frames
// enqueued on ONE stream follow each other; a frame on another stream would run beside the previous
// one and both would write the same lists, so it is refused until the previous frame has completed.
// (Under stream capture nothing is enqueued yet and events cannot be queried: the caller who replays
// captured frames keeps them apart himself.)
static int noalloc_flight_check( int device, hipStream_t s, bool &capturing )
{
	NoallocFlight &f = g_flight[device];
	hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
	HIPCHK( hipStreamIsCapturing( s, &cs ) );
	capturing = cs != hipStreamCaptureStatusNone;
	if ( capturing || !f.pending || f.stream == s )
		return 0;
	const hipError_t q = hipEventQuery( f.done );
	if ( q == hipErrorNotReady )
		return fail( "rt_hip_render_noalloc: the previous allocation-free frame on GPU %d, enqueued on another stream, has not completed; "
			"the entry point is single-flight per GPU (one shared work area): synchronise that stream first, or enqueue on the same stream", device );
	HIPCHK( q );
	f.pending = false;
	return 0;
}

int rt_hip_render_noalloc( rt_hip_ctx *c, void *d_out, unsigned flags )
{
	if ( !c ) return fail( "null context" );
	if ( !d_out ) return fail( "rt_hip_render_noalloc: an output buffer is required" );
	if ( flags != 0u ) return fail( "rt_hip_render_noalloc: flags are not supported (counting, hit ids and timing use buffers of rt_hip_render)" );
	if ( lender_unchanged( c, "rt_hip_render_noalloc" ) ) return 1;
	if ( !c->S.nodes || !c->S.mats || !c->S.spheres || !c->S.lights )
		return fail( "rt_hip_render_noalloc: upload tree, materials, spheres and lights first (empty tables included)" );
	if ( ( c->nRecs && c->maxMatTree >= c->nMats ) || ( c->S.nSpheres && c->maxMatSphere >= c->nMats ) )
		return fail( "rt_hip_render_noalloc: geometry references a material slot beyond the uploaded table" );
	const DevScene &S = c->S;
	if ( S.w < 2 || S.h < 2 ) return fail( "rt_hip_render_noalloc: image size not set" );
	const size_t npix = (size_t) S.w * (size_t) S.localH;
	if ( npix > RT_HIP_NOALLOC_MAX_PIXELS )
		return fail( "rt_hip_render_noalloc: %zu pixels exceed the limit of %d (use rt_hip_render, or render in tiles)", npix, RT_HIP_NOALLOC_MAX_PIXELS );
	if ( S.stackSize > RT_HIP_NOALLOC_MAX_SLOTS )
		return fail( "rt_hip_render_noalloc: %d rays per pixel exceed the limit of %d", S.stackSize, RT_HIP_NOALLOC_MAX_SLOTS );
	if ( S.nLights > RT_HIP_NOALLOC_MAX_LIGHTS )
		return fail( "rt_hip_render_noalloc: %d lights exceed the limit of %d", S.nLights, RT_HIP_NOALLOC_MAX_LIGHTS );
	HIPCHK( hipSetDevice( c->device ) );
	unsigned char *base = g_scratch_ptr[c->device];
	if ( !base ) return fail( "rt_hip_render_noalloc: no scratch address for device %d", c->device );
	bool capturing = false;
	if ( noalloc_flight_check( c->device, c->stream, capturing ) ) return 1;

	typedef NoallocLayout L;
	Work W;
	memset( &W, 0, sizeof( W ) );
	const size_t per = L::cap * ( L::slots - 1 );
	W.ctr = (Counters *) ( base + L::oCtr );
	W.rayA = (float4 *) ( base + L::oRays );
	W.rayB = W.rayA + per;
	W.rayC = W.rayB + per;
	W.rayD = W.rayC + per;
	W.hitA = (float4 *) ( base + L::oHitA );
	W.hitB = (uint2 *) ( base + L::oHitB );
	W.hitN = (float4 *) ( base + L::oHitN );
	W.res = (float4 *) ( base + L::oRes );
	W.fac = (float4 *) ( base + L::oFac );
	W.shadow = (float *) ( base + L::oShadow );
	W.stackPos = (uint32_t *) ( base + L::oStackPos );
	W.hitId = (int32_t *) ( base + L::oHitId );
	W.spill = (uint2 *) ( base + L::oSpill );
	W.npix = (uint32_t) npix;
	W.cap = (uint32_t) L::cap;
	W.tilesX = (uint32_t) ( ( S.w + 7 ) / 8 );
	W.nChunks0 = W.tilesX * (uint32_t) ( ( S.localH + 7 ) / 8 );
	W.spillStride = (uint32_t) RT_NOALLOC_GRID * RT_BLOCK;
	W.exactAppends = 1u;
	// one stream, passes in order: one stack-overflow strip is enough
	if ( enqueue_frame( c, W, (rt_argb *) d_out, 0u, false, RT_NOALLOC_GRID ) ) return 1;
	if ( !capturing )
	{
		NoallocFlight &f = g_flight[c->device];
		HIPCHK( hipEventRecord( f.done, c->stream ) );
		f.stream = c->stream;
		f.pending = true;
	}
	return 0;
}

// after a synchronisation: did any list append of the last frame run out of room?
static int check_overflow( rt_hip_ctx *c )
{
	if ( !c->lastCtr )
		return 0;
	uint32_t ov = 0;
	HIPCHK( hipMemcpyAsync( &ov, (char *) c->lastCtr + offsetof( Counters, overflow ), sizeof( ov ), hipMemcpyDeviceToHost, c->stream ) );
	HIPCHK( hipStreamSynchronize( c->stream ) );
	if ( ov )
		return fail( "the last frame overflowed work list %u (1 hits, 2 rays; capacity %u entries): the image is incomplete", ov, c->lastCap );
	return 0;
}

int rt_hip_read_frame( rt_hip_ctx *c, rt_argb *host )
{
	if ( !c || !host ) return fail( "rt_hip_read_frame: bad arguments" );
	HIPCHK( hipSetDevice( c->device ) );
	if ( !c->frame.p ) return fail( "rt_hip_read_frame: nothing rendered" );
	HIPCHK( hipMemcpyAsync( host, c->frame.p, sizeof( rt_argb ) * (size_t) c->W.npix, hipMemcpyDeviceToHost, c->stream ) );
	HIPCHK( hipStreamSynchronize( c->stream ) );
	return check_overflow( c );
}

void *rt_hip_alloc_host( rt_hip_ctx *c, size_t bytes )
{
	void *p = nullptr;
	if ( !c || !bytes || hipSetDevice( c->device ) != hipSuccess || hipHostMalloc( &p, bytes, hipHostMallocDefault ) != hipSuccess )
	{
		// the caller falls back to pageable memory: do not leave the failure behind
		// as the runtime's last error, where the next launch check would find it
		(void) hipGetLastError();
		fail( "rt_hip_alloc_host: %zu bytes of page-locked memory not available", bytes );
		return nullptr;
	}
	return p;
}

void rt_hip_free_host( rt_hip_ctx *c, void *host )
{
	if ( !c || !host )
		return;
	(void) hipSetDevice( c->device );
	(void) hipHostFree( host );
}

int rt_hip_read_hit_ids( rt_hip_ctx *c, int32_t *host )
{
	if ( !c || !host ) return fail( "rt_hip_read_hit_ids: bad arguments" );
	HIPCHK( hipSetDevice( c->device ) );
	if ( !c->hitId.p ) return fail( "rt_hip_read_hit_ids: nothing rendered" );
	HIPCHK( hipMemcpyAsync( host, c->hitId.p, sizeof( int32_t ) * (size_t) c->W.npix, hipMemcpyDeviceToHost, c->stream ) );
	HIPCHK( hipStreamSynchronize( c->stream ) );
	return 0;
}

int rt_hip_get_stats( rt_hip_ctx *c, rt_hip_stats *out )
{
	if ( !c || !out ) return fail( "rt_hip_get_stats: bad arguments" );
	HIPCHK( hipSetDevice( c->device ) );
	if ( !c->ctr.p ) return fail( "rt_hip_get_stats: nothing rendered" );
	// list lengths include reservation holes: count the valid entries
	Counters *d = (Counters *) c->ctr.p;
	HIPCHK( hipMemsetAsync( d->validRays, 0, sizeof( d->validRays ) + sizeof( d->validHits ), c->stream ) );
	for ( int k = 0; k < c->S.stackSize; ++k )
	{
		if ( k > 0 )
			hipLaunchKernelGGL( k_count_valid, dim3( 1024 ), dim3( RT_BLOCK ), 0, c->stream,
				(const float4 *) c->W.rayA + (size_t) ( k - 1 ) * c->W.cap, (const uint32_t *) &d->rayCount[k].v, 3, &d->validRays[k] );
		hipLaunchKernelGGL( k_count_valid, dim3( 1024 ), dim3( RT_BLOCK ), 0, c->stream,
			(const float4 *) c->W.hitA + (size_t) k * c->W.cap, (const uint32_t *) &d->hitCount[k].v, 0, &d->validHits[k] );
	}
	Counters h;
	HIPCHK( hipMemcpyAsync( &h, c->ctr.p, sizeof( h ), hipMemcpyDeviceToHost, c->stream ) );
	HIPCHK( hipStreamSynchronize( c->stream ) );
	if ( h.overflow )
		return fail( "the last frame overflowed work list %u (1 hits, 2 rays; capacity %u entries): the counters are incomplete; rays %u %u %u %u, hits %u %u %u %u %u", h.overflow, c->W.cap,
			h.rayCount[1].v, h.rayCount[2].v, h.rayCount[3].v, h.rayCount[4].v,
			h.hitCount[0].v, h.hitCount[1].v, h.hitCount[2].v, h.hitCount[3].v, h.hitCount[4].v );
#ifdef RT_PROFILE_CHUNKS
	for ( int k = 0; k < c->S.stackSize; ++k )
	{
		fprintf( stderr, "[prof] slot %d: %llu chunks, per chunk %.1f lane-per-ray rounds (max %u), %.1f cooperative rounds (max %u), %.1f expand steps (max %u); duration histogram (10 us bins):",
			k, h.chunksSeen[k], (double) h.lockRounds[k] / ( h.chunksSeen[k] ? h.chunksSeen[k] : 1 ), h.maxLock[k],
			(double) h.coopRounds[k] / ( h.chunksSeen[k] ? h.chunksSeen[k] : 1 ), h.maxCoop[k],
			(double) h.coopIters[k] / ( h.chunksSeen[k] ? h.chunksSeen[k] : 1 ), h.maxIters[k] );
		const double n = h.chunksSeen[k] ? (double) h.chunksSeen[k] : 1.0;
		fprintf( stderr, " [us per chunk: total %.1f, cooperative %.1f of which expand %.1f scan %.1f]", h.tChunk[k] / n / 100.0, h.tCoop[k] / n / 100.0, h.tExpand[k] / n / 100.0, h.tScan[k] / n / 100.0 );
		fprintf( stderr, " [lane-per-ray: descend %.1f leaf %.1f; finish %.1f setup %.1f]", h.tDescend[k] / n / 100.0, h.tLeaf[k] / n / 100.0, h.tFinish[k] / n / 100.0, h.tSetup[k] / n / 100.0 );
		fprintf( stderr, " [direction octants per chunk %.2f, rays in the chunk's most frequent octant %.3f]", h.octDistinct[k] / n, h.octRays[k] ? (double) h.octMajor[k] / (double) h.octRays[k] : 0.0 );
		if ( getenv( "RT_PROF_HIST" ) ) for ( int b = 0; b < 64; ++b ) if ( h.chunkHist[k][b] ) fprintf( stderr, " %d:%u", b, h.chunkHist[k][b] );
		fprintf( stderr, "\n" );
	}
#endif
	memset( out, 0, sizeof( *out ) );
	out->raysClosest = c->validPixels;
	for ( int k = 1; k < RT_MAX_SLOTS; ++k ) out->raysClosest += h.validRays[k];
	for ( int k = 0; k < RT_MAX_SLOTS; ++k ) out->hits += h.validHits[k];
	out->raysShadow = out->hits * (uint64_t) c->S.nLights;
	out->nodeVisits = h.nodeVisits;
	out->triTests = h.triTests;
	out->spillPushes = h.spillPushes;
	out->maxStack = h.maxStack;
	out->nodeVisitsShadow = h.nodeVisitsShadow;
	out->triTestsShadow = h.triTestsShadow;
	out->nodeSlots = h.nodeSlots;
	out->triSlots = h.triSlots;
	out->nodeSlotsShadow = h.nodeSlotsShadow;
	out->triSlotsShadow = h.triSlotsShadow;
	for ( int k = 0; k < RT_MAX_SLOTS; ++k )
	{
		out->slotRays[k] = k ? h.validRays[k] : c->validPixels;
		out->slotHits[k] = h.validHits[k];
	}
	return 0;
}

int rt_hip_selftest_rcp( rt_hip_ctx *c, uint64_t *mismatches, uint64_t *fastPath )
{
	if ( !c || !mismatches || !fastPath ) return fail( "rt_hip_selftest_rcp: bad arguments" );
	HIPCHK( hipSetDevice( c->device ) );
	unsigned long long *d = nullptr, h[2] = { 0, 0 };
	HIPCHK( hipMalloc( &d, sizeof( h ) ) );
	HIPCHK( hipMemsetAsync( d, 0, sizeof( h ), c->stream ) );
	hipLaunchKernelGGL( k_selftest_rcp, dim3( 4096 ), dim3( RT_BLOCK ), 0, c->stream, d, d + 1 );
	hipError_t e = hipMemcpyAsync( h, d, sizeof( h ), hipMemcpyDeviceToHost, c->stream );
	if ( e == hipSuccess ) e = hipStreamSynchronize( c->stream );
	(void) hipFree( d );
	HIPCHK( e );
	*mismatches = h[0];
	*fastPath = h[1];
	return 0;
}

int rt_hip_kernel_times( rt_hip_ctx *c, float ms[5], int launches[5] )
{
	return rt_hip_kernel_times_ex( c, ms, launches, nullptr );
}

// Launch intervals of one timed frame in ms after the frame's own start event.  Only the end of
// every kernel is recorded (see enqueue_frame); a kernel starts when the last of its predecessors
// has ended: the kernel before it on its stream and, where its stream waited for an event, the
// kernel that event follows.  a[j][k] .. b[j][k]: kernel j (0 closest-hit, 1 shadow, 2 lighting) of
// slot k; fold[2], frame end.  (The gap between a predecessor's end and the actual start of the
// next kernel - a few microseconds of dispatch - is counted as the kernel's.)
struct FrameTimes { float a[3][RT_MAX_SLOTS], b[3][RT_MAX_SLOTS], fold[2], end; };
static int frame_times( rt_hip_ctx *c, int slotOfRing, FrameTimes &ft )
{
	hipEvent_t *ev = c->ev[slotOfRing];
	const bool overlap = c->evOverlap[slotOfRing];
	const int n = c->timedSlots;
	hipEvent_t zero = ev[6 * RT_MAX_SLOTS + 2];
	auto at = [&]( hipEvent_t e, float &t ) -> int { HIPCHK( hipEventElapsedTime( &t, zero, e ) ); return 0; };
	float first = 0.0f;
	if ( at( ev[0], first ) ) return 1;
	for ( int k = 0; k < n; ++k )
		for ( int j = 0; j < 3; ++j )
			if ( at( ev[6 * k + 2 * j + 1], ft.b[j][k] ) ) return 1;
	if ( at( ev[6 * RT_MAX_SLOTS + 3], ft.end ) ) return 1;
	auto later = []( float x, float y ) { return x > y ? x : y; };
	for ( int k = 0; k < n; ++k )
	{
		if ( overlap )
		{
			// chain on the caller's stream; shadow / lighting of slot k on auxiliary stream k % RT_AUX_STREAMS
			ft.a[0][k] = k ? ft.b[0][k - 1] : first;
			ft.a[1][k] = k >= RT_AUX_STREAMS ? later( ft.b[0][k], ft.b[2][k - RT_AUX_STREAMS] ) : ft.b[0][k];
		}
		else
		{
			ft.a[0][k] = k ? ft.b[2][k - 1] : first;
			ft.a[1][k] = ft.b[0][k];
		}
		ft.a[2][k] = ft.b[1][k];
	}
	ft.fold[0] = n ? ft.b[0][n - 1] : first;
	for ( int k = 0; k < n; ++k )
		ft.fold[0] = later( ft.fold[0], ft.b[2][k] );
	ft.fold[1] = ft.end;
	return 0;
}

// Every launch of the frames rendered with RT_HIP_TIMING since the last read-out, as an interval
// [start, end] in milliseconds after the start of the oldest such frame of `origin` (a context on
// the same GPU; `c` itself when null): contexts that render frames side by side are brought onto
// one clock, so a caller can take the union of a kernel's intervals over all of them.
// kind: 0 closest, 1 shadow, 2 light, 3 fold, 4 whole frame.  Does not reset the timings.
int rt_hip_kernel_intervals( rt_hip_ctx *c, rt_hip_ctx *origin, float *startEnd, int *kind, int cap, int *count )
{
	if ( !c || !startEnd || !kind || !count || cap < 0 ) return fail( "rt_hip_kernel_intervals: bad arguments" );
	if ( !origin ) origin = c;
	if ( c->device != origin->device ) return fail( "rt_hip_kernel_intervals: contexts on different devices" );
	if ( c->timedFrames == 0 || origin->timedFrames == 0 )
		return fail( "rt_hip_kernel_intervals: no frame was rendered with RT_HIP_TIMING since the last read-out" );
	HIPCHK( hipSetDevice( c->device ) );
	HIPCHK( hipStreamSynchronize( c->stream ) );
	HIPCHK( hipStreamSynchronize( origin->stream ) );
	const int kept = origin->timedFrames < RT_TIMED_FRAMES ? origin->timedFrames : RT_TIMED_FRAMES;
	hipEvent_t zero = origin->ev[( origin->timedFrames - kept ) % RT_TIMED_FRAMES][6 * RT_MAX_SLOTS + 2];
	const int n = c->timedSlots;
	const int frames = c->timedFrames < RT_TIMED_FRAMES ? c->timedFrames : RT_TIMED_FRAMES;
	if ( frames * ( 3 * n + 2 ) > cap ) return fail( "rt_hip_kernel_intervals: room for %d intervals, %d needed", cap, frames * ( 3 * n + 2 ) );
	int m = 0;
	auto put = [&]( float a, float b, int what ) { startEnd[2 * m] = a; startEnd[2 * m + 1] = b; kind[m] = what; ++m; };
	for ( int f = frames - 1; f >= 0; --f )
	{
		const int ring = ( c->timedFrames - 1 - f ) % RT_TIMED_FRAMES;
		FrameTimes ft;
		if ( frame_times( c, ring, ft ) ) return 1;
		float t0 = 0.0f;
		HIPCHK( hipEventElapsedTime( &t0, zero, c->ev[ring][6 * RT_MAX_SLOTS + 2] ) );
		for ( int k = 0; k < n; ++k )
			for ( int j = 0; j < 3; ++j )
				put( t0 + ft.a[j][k], t0 + ft.b[j][k], j );
		put( t0 + ft.fold[0], t0 + ft.fold[1], 3 );
		put( t0, t0 + ft.end, 4 );
	}
	*count = m;
	return 0;
}

int rt_hip_kernel_times_ex( rt_hip_ctx *c, float ms[5], int launches[5], float wall[4] )
{
	if ( !c || !ms || !launches ) return fail( "rt_hip_kernel_times: bad arguments" );
	if ( c->timedFrames == 0 ) return fail( "rt_hip_kernel_times: no frame was rendered with RT_HIP_TIMING since the last call" );
	HIPCHK( hipSetDevice( c->device ) );
	HIPCHK( hipStreamSynchronize( c->stream ) );
	for ( int i = 0; i < 5; ++i ) ms[i] = 0.0f;
	if ( wall ) for ( int i = 0; i < 4; ++i ) wall[i] = 0.0f;
	const int n = c->timedSlots;
	const int frames = c->timedFrames < RT_TIMED_FRAMES ? c->timedFrames : RT_TIMED_FRAMES;
	for ( int f = 0; f < frames; ++f )
	{
		FrameTimes ft;
		if ( frame_times( c, ( c->timedFrames - 1 - f ) % RT_TIMED_FRAMES, ft ) ) return 1;
		for ( int j = 0; j < 3; ++j )             // closest, shadow, light
		{
			float a[RT_MAX_SLOTS], b[RT_MAX_SLOTS];
			for ( int k = 0; k < n; ++k )
			{
				a[k] = ft.a[j][k];
				b[k] = ft.b[j][k];
				ms[j] += b[k] - a[k];
			}
			if ( wall )
			{
				// length of the union of the intervals: the wall-clock time during which
				// at least one launch of this kernel was running
				for ( int x = 1; x < n; ++x )
					for ( int y = x; y > 0 && a[y] < a[y - 1]; --y )
					{
						float ta = a[y], tb = b[y];
						a[y] = a[y - 1]; b[y] = b[y - 1];
						a[y - 1] = ta; b[y - 1] = tb;
					}
				float end = -1.0f;
				for ( int k = 0; k < n; ++k )
				{
					const float lo = a[k] > end ? a[k] : end;
					if ( b[k] > lo ) { wall[j] += b[k] - lo; end = b[k]; }
				}
			}
		}
		ms[3] += ft.fold[1] - ft.fold[0];
		if ( wall ) wall[3] += ft.fold[1] - ft.fold[0];
		ms[4] += ft.end;
	}
	for ( int i = 0; i < 5; ++i ) ms[i] /= (float) frames;
	if ( wall ) for ( int i = 0; i < 4; ++i ) wall[i] /= (float) frames;
	launches[0] = launches[1] = launches[2] = n;
	launches[3] = 1;
	launches[4] = 3 * n + 1;
	c->timedFrames = 0;
	return 0;
}
