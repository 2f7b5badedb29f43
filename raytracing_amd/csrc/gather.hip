// Multi-GPU frame assembly behind the C-ABI (gfx950 only).
//
// The reference renders a frame in one NDRange on one device
// (rt_funcs_render_pipe.c:534-569) and has no counterpart for any of this.
// Here the image is cut into bands of bandH rows dealt round-robin to the
// ranks (rt_hip_set_tile): rank r renders bands r, r + N, ... and stores them
// compactly.  Assembling the frame is one exchange step:
//
//   every rank sends its compact share, padded to the same number of whole
//   bands, straight to the destination rank (RCCL send / receive inside one
//   group: all-to-one, every source uses its own xGMI link to the
//   destination; 4K: 33 MB per frame in total), and the destination
//   de-interleaves the N staged shares into the frame with one kernel (local
//   band b of rank r is band b * N + r of the image).
//
// RCCL is opened at run time (dlopen): the library has no load-time dependency
// on it, and a process that already has RCCL mapped (PyTorch's copy) keeps
// using that one.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "rt_hip.h"

namespace {

thread_local char g_err[512] = "";

int fail( const char *fmt, ... )
{
	va_list ap;
	va_start( ap, fmt );
	vsnprintf( g_err, sizeof( g_err ), fmt, ap );
	va_end( ap );
	return 1;
}

#define HIPCHK( x ) do { hipError_t e_ = ( x ); if ( e_ != hipSuccess ) \
	return fail( "%s failed: %s (%s:%d)", #x, hipGetErrorString( e_ ), __FILE__, __LINE__ ); } while ( 0 )

struct Rccl
{
	void *so = nullptr;
	ncclResult_t ( *GetUniqueId )( ncclUniqueId * ) = nullptr;
	ncclResult_t ( *CommInitRank )( ncclComm_t *, int, ncclUniqueId, int ) = nullptr;
	ncclResult_t ( *CommDestroy )( ncclComm_t ) = nullptr;
	ncclResult_t ( *GroupStart )() = nullptr;
	ncclResult_t ( *GroupEnd )() = nullptr;
	ncclResult_t ( *Send )( const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t ) = nullptr;
	ncclResult_t ( *Recv )( void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t ) = nullptr;
	const char *( *GetErrorString )( ncclResult_t ) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int load_rccl()
{
	// (several host threads may open communicators at once)
	std::lock_guard<std::mutex> lock( g_rccl_mu );
	if ( g_rccl.so )
		return 0;
	const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
	void *so = nullptr;
	for ( const char *n : names )
		if ( ( so = dlopen( n, RTLD_NOW | RTLD_GLOBAL ) ) )
			break;
	if ( !so )
		return fail( "RCCL not found (librccl.so.1): %s", dlerror() );
	Rccl r;
	r.so = so;
	bool ok = true;
	auto sym = [&]( const char *name ) { void *p = dlsym( so, name ); ok = ok && p; return p; };
	r.GetUniqueId = ( decltype( r.GetUniqueId ) ) sym( "ncclGetUniqueId" );
	r.CommInitRank = ( decltype( r.CommInitRank ) ) sym( "ncclCommInitRank" );
	r.CommDestroy = ( decltype( r.CommDestroy ) ) sym( "ncclCommDestroy" );
	r.GroupStart = ( decltype( r.GroupStart ) ) sym( "ncclGroupStart" );
	r.GroupEnd = ( decltype( r.GroupEnd ) ) sym( "ncclGroupEnd" );
	r.Send = ( decltype( r.Send ) ) sym( "ncclSend" );
	r.Recv = ( decltype( r.Recv ) ) sym( "ncclRecv" );
	r.GetErrorString = ( decltype( r.GetErrorString ) ) sym( "ncclGetErrorString" );
	if ( !ok )
		return fail( "RCCL library lacks a required symbol" );
	g_rccl = r;
	return 0;
}

#define NCCLCHK( x ) do { ncclResult_t r_ = ( x ); if ( r_ != ncclSuccess ) \
	return fail( "%s failed: %s", #x, g_rccl.GetErrorString( r_ ) ); } while ( 0 )
// inside an open group: the group is closed (whatever that returns) before the error is reported,
// so the thread's later RCCL calls do not land in a group nobody ends
#define NCCLCHK_IN_GROUP( x ) do { ncclResult_t r_ = ( x ); if ( r_ != ncclSuccess ) { \
	(void) g_rccl.GroupEnd(); return fail( "%s failed: %s", #x, g_rccl.GetErrorString( r_ ) ); } } while ( 0 )

// out[y][x] = stage[band % nParts][(band / nParts) * bandH + y % bandH][x], band = y / bandH; one 4-byte pixel per thread
__global__ __launch_bounds__( 256 ) void k_deinterleave( const uint32_t *stage, uint32_t *out, int w, int h, int bandH, int nParts, int padRows )
{
	const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
	if ( i >= (size_t) w * h )
		return;
	const int y = (int) ( i / (size_t) w ), x = (int) ( i - (size_t) y * w );
	const int band = y / bandH, r = band % nParts, row = ( band / nParts ) * bandH + ( y - band * bandH );
	out[i] = stage[( (size_t) r * padRows + row ) * w + x];
}

int padded_rows( int h, int bandH, int nParts )
{
	const int bands = ( h + bandH - 1 ) / bandH;
	return ( ( bands + nParts - 1 ) / nParts ) * bandH;
}

} // namespace

struct rt_hip_comm
{
	int device = 0, nRanks = 1, rank = 0;
	ncclComm_t comm = nullptr;
	void *stage = nullptr;          // destination rank: nRanks padded shares
	size_t stageBytes = 0;
	hipStream_t lastStream = nullptr;   // stream of the gather enqueued last (it may still be writing `stage`)
	bool used = false;
};

extern "C" {

const char *rt_hip_comm_last_error( void ) { return g_err; }

int rt_hip_comm_unique_id( void *id128 )
{
	if ( !id128 ) return fail( "rt_hip_comm_unique_id: null argument" );
	if ( load_rccl() ) return 1;
	static_assert( sizeof( ncclUniqueId ) == RT_HIP_COMM_ID_BYTES, "RT_HIP_COMM_ID_BYTES must be the size of an RCCL unique id" );
	ncclUniqueId id;
	NCCLCHK( g_rccl.GetUniqueId( &id ) );
	memcpy( id128, &id, sizeof( id ) );
	return 0;
}

rt_hip_comm *rt_hip_comm_create( int device, int nRanks, int rank, const void *id128 )
{
	if ( nRanks < 1 || rank < 0 || rank >= nRanks || !id128 )
	{
		fail( "rt_hip_comm_create: bad arguments" );
		return nullptr;
	}
	if ( load_rccl() )
		return nullptr;
	if ( hipSetDevice( device ) != hipSuccess )
	{
		fail( "rt_hip_comm_create: hipSetDevice(%d) failed", device );
		return nullptr;
	}
	rt_hip_comm *m = new rt_hip_comm();
	m->device = device;
	m->nRanks = nRanks;
	m->rank = rank;
	ncclUniqueId id;
	memcpy( &id, id128, sizeof( id ) );
	ncclResult_t r = g_rccl.CommInitRank( &m->comm, nRanks, id, rank );
	if ( r != ncclSuccess )
	{
		fail( "ncclCommInitRank failed: %s", g_rccl.GetErrorString( r ) );
		delete m;
		return nullptr;
	}
	return m;
}

void rt_hip_comm_destroy( rt_hip_comm *m )
{
	if ( !m )
		return;
	(void) hipSetDevice( m->device );
	if ( m->stage ) (void) hipFree( m->stage );
	if ( m->comm ) (void) g_rccl.CommDestroy( m->comm );
	delete m;
}

int rt_hip_deinterleave( const void *d_stage, void *d_frame, int w, int h, int bandH, int nParts, void *hipStream )
{
	if ( !d_stage || !d_frame || w < 1 || h < 1 || bandH < 1 || nParts < 1 ) return fail( "rt_hip_deinterleave: bad arguments" );
	const size_t n = (size_t) w * h;
	hipLaunchKernelGGL( k_deinterleave, dim3( (unsigned) ( ( n + 255 ) / 256 ) ), dim3( 256 ), 0, (hipStream_t) hipStream,
		(const uint32_t *) d_stage, (uint32_t *) d_frame, w, h, bandH, nParts, padded_rows( h, bandH, nParts ) );
	HIPCHK( hipGetLastError() );
	return 0;
}

int rt_hip_gather_rows( int h, int bandH, int nParts ) { return ( h < 1 || bandH < 1 || nParts < 1 ) ? 0 : padded_rows( h, bandH, nParts ); }

int rt_hip_gather_frame( rt_hip_comm *m, const void *d_local, void *d_frame, int w, int h, int bandH, int dst, void *hipStream )
{
	if ( !m || !d_local || w < 1 || h < 1 || bandH < 1 || dst < 0 || dst >= m->nRanks ) return fail( "rt_hip_gather_frame: bad arguments" );
	if ( m->rank == dst && !d_frame ) return fail( "rt_hip_gather_frame: the destination rank needs a frame buffer" );
	hipStream_t s = (hipStream_t) hipStream;
	HIPCHK( hipSetDevice( m->device ) );
	const int pad = padded_rows( h, bandH, m->nRanks );
	const size_t slab = (size_t) pad * w * sizeof( rt_argb );
	if ( m->rank == dst && m->stageBytes < slab * m->nRanks )
	{
		// (first frame, or a larger image than before: the only allocation of the exchange)
		HIPCHK( hipStreamSynchronize( s ) );
		// a gather enqueued earlier on another stream may still be filling the old stage
		if ( m->used && m->lastStream != s ) HIPCHK( hipStreamSynchronize( m->lastStream ) );
		if ( m->stage ) HIPCHK( hipFree( m->stage ) );
		m->stage = nullptr;
		m->stageBytes = 0;
		HIPCHK( hipMalloc( &m->stage, slab * m->nRanks ) );
		m->stageBytes = slab * m->nRanks;
	}
	m->lastStream = s;
	m->used = true;
	if ( m->rank == dst )
	{
		// own share: a device copy; the others: one receive each, all inside one group
		HIPCHK( hipMemcpyAsync( (char *) m->stage + slab * dst, d_local, slab, hipMemcpyDeviceToDevice, s ) );
		if ( m->nRanks > 1 )
		{
			NCCLCHK( g_rccl.GroupStart() );
			for ( int r = 0; r < m->nRanks; ++r )
				if ( r != dst )
					NCCLCHK_IN_GROUP( g_rccl.Recv( (char *) m->stage + slab * r, slab, ncclUint8, r, m->comm, s ) );
			NCCLCHK( g_rccl.GroupEnd() );
		}
		return rt_hip_deinterleave( m->stage, d_frame, w, h, bandH, m->nRanks, s );
	}
	NCCLCHK( g_rccl.GroupStart() );
	NCCLCHK_IN_GROUP( g_rccl.Send( d_local, slab, ncclUint8, dst, m->comm, s ) );
	NCCLCHK( g_rccl.GroupEnd() );
	return 0;
}

} // extern "C"
