// Host vector / matrix / colour helpers behind include/rt_funcs_math.h.
// Results match the reference's rt_funcs_math.c function for function
// (same fp32 operand order; trigonometry evaluated in double and rounded on
// store, as C's promotion rules make the reference do).
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "rt_funcs_math.h"

namespace {

inline float *M( rt_matrix4 *m ) { return &m->_11; }
inline const float *M( const rt_matrix4 *m ) { return &m->_11; }

void set_rows( rt_matrix4 *m, const float r[16] ) { memcpy( M( m ), r, 64 ); }

// row-vector product restricted to the upper-left dim x dim block:
// out[c] = sum_r v[r] * m[r][c] (+ translation row when affine)
void vec_mat( const rt_vector3 *v, const rt_matrix4 *m, bool affine, rt_vector3 *out )
{
	const float *a = M( m );
	rt_vector3 t;
	t.x = a[0] * v->x + a[4] * v->y + a[8] * v->z;
	t.y = a[1] * v->x + a[5] * v->y + a[9] * v->z;
	t.z = a[2] * v->x + a[6] * v->y + a[10] * v->z;
	if ( affine )
	{
		t.x = t.x + a[12];
		t.y = t.y + a[13];
		t.z = t.z + a[14];
	}
	t._pad = out->_pad;
	*out = t;
}

// R = M0 * M1 (reference rt_funcs_math.c:410-451; every element is a
// left-to-right sum of four products, second factor taken from M0)
void mat_mult4( const rt_matrix4 *m0, const rt_matrix4 *m1, rt_matrix4 *out )
{
	const float *a = M( m0 ), *b = M( m1 );
	float r[16];
	for ( int i = 0; i < 4; ++i )
		for ( int j = 0; j < 4; ++j )
		{
			float s = b[j] * a[4 * i];
			for ( int k = 1; k < 4; ++k )
				s = s + b[4 * k + j] * a[4 * i + k];
			r[4 * i + j] = s;
		}
	memcpy( M( out ), r, 64 );
}

// Gauss-Jordan without pivoting, as the reference (rt_funcs_math.c:312-398)
void gauss_jordan( const float *src, int dim, float *dst )
{
	float t[4][4], inv[4][4];
	for ( int i = 0; i < dim; ++i )
		for ( int j = 0; j < dim; ++j )
		{
			t[i][j] = src[dim * i + j];
			inv[i][j] = ( i == j ) ? 1.0f : 0.0f;
		}
	for ( int k = 0; k < dim; ++k )
	{
		float coef = 1 / t[k][k];
		for ( int j = 0; j < dim; ++j )
		{
			t[k][j] *= coef;
			inv[k][j] *= coef;
		}
		for ( int i = 0; i < dim; ++i )
			if ( i != k )
			{
				coef = -( t[i][k] / t[k][k] );
				for ( int j = 0; j < dim; ++j )
				{
					t[i][j] += t[k][j] * coef;
					inv[i][j] += inv[k][j] * coef;
				}
			}
	}
	for ( int i = 0; i < dim; ++i )
		for ( int j = 0; j < dim; ++j )
			dst[dim * i + j] = inv[i][j];
}

} // namespace

extern "C" {

void rt_vector3_create( rt_vector3 *pV, rt_float x, rt_float y, rt_float z )
{
	if ( !pV ) exit( -1 );
	pV->x = x; pV->y = y; pV->z = z;
}

void rt_color_create( rt_color *pC, rt_float a, rt_float r, rt_float g, rt_float b )
{
	if ( !pC ) exit( -1 );
	pC->a = a; pC->r = r; pC->g = g; pC->b = b;
}

// The reference's rt_matrix3 is the 4x4 struct; its 3x3 constructors touch the
// nine named members _11.._33 only.
void rt_matrix3_create_scale( rt_matrix3 *pR, rt_float x, rt_float y, rt_float z )
{
	if ( !pR ) exit( -1 );
	pR->_11 = x; pR->_12 = 0; pR->_13 = 0;
	pR->_21 = 0; pR->_22 = y; pR->_23 = 0;
	pR->_31 = 0; pR->_32 = 0; pR->_33 = z;
}

void rt_matrix3_create_rotate( rt_matrix3 *pR, rt_float ang, RT_AXIS a )
{
	if ( !pR ) exit( -1 );
	// the reference is C: cos / sin / tan of a float are the DOUBLE functions there, rounded on assignment
	const float c = (float) cos( (double) ang ), s = (float) sin( (double) ang ), ns = (float) -sin( (double) ang );
	if ( a == RT_AXIS_X )
	{
		pR->_11 = 1; pR->_12 = 0; pR->_13 = 0;
		pR->_21 = 0; pR->_22 = c; pR->_23 = s;
		pR->_31 = 0; pR->_32 = ns; pR->_33 = c;
	}
	else if ( a == RT_AXIS_Y )
	{
		pR->_11 = c; pR->_12 = 0; pR->_13 = ns;
		pR->_21 = 0; pR->_22 = 1; pR->_23 = 0;
		pR->_31 = s; pR->_32 = 0; pR->_33 = c;
	}
}

void rt_matrix4_create_translate( rt_matrix4 *pR, rt_float x, rt_float y, rt_float z )
{
	if ( !pR ) exit( -1 );
	const float r[16] = { 1, 0, 0, 0,  0, 1, 0, 0,  0, 0, 1, 0,  x, y, z, 1 };
	set_rows( pR, r );
}

void rt_matrix4_create_scale( rt_matrix4 *pR, rt_float x, rt_float y, rt_float z )
{
	if ( !pR ) exit( -1 );
	const float r[16] = { x, 0, 0, 0,  0, y, 0, 0,  0, 0, z, 0,  0, 0, 0, 1 };
	set_rows( pR, r );
}

void rt_matrix4_create_rotate( rt_matrix4 *pR, rt_float ang, RT_AXIS a )
{
	if ( !pR ) exit( -1 );
	// the reference is C: cos / sin / tan of a float are the DOUBLE functions there, rounded on assignment
	const float c = (float) cos( (double) ang ), s = (float) sin( (double) ang ), ns = (float) -sin( (double) ang );
	if ( a == RT_AXIS_X )
	{
		const float r[16] = { 1, 0, 0, 0,  0, c, s, 0,  0, ns, c, 0,  0, 0, 0, 1 };
		set_rows( pR, r );
	}
	else if ( a == RT_AXIS_Y )
	{
		const float r[16] = { c, 0, ns, 0,  0, 1, 0, 0,  s, 0, c, 0,  0, 0, 0, 1 };
		set_rows( pR, r );
	}
	else if ( a == RT_AXIS_Z )
	{
		const float r[16] = { c, s, 0, 0,  ns, c, 0, 0,  0, 0, 1, 0,  0, 0, 0, 1 };
		set_rows( pR, r );
	}
}

// maps NDC x,y in [-1,1] to the view plane at z = 1 (reference :127-137)
void rt_matrix4_create_projection( rt_matrix4 *pR, rt_float asp, rt_float hV )
{
	// C semantics: tan( double ), products in double, one rounding on assignment
	const double t = tan( (double) ( hV / 2.0f ) );
	const float sx = (float) ( 2.0 * t * (double) asp );
	const float sy = (float) ( 2.0 * t );
	const float r[16] = { sx, 0, 0, 0,  0, sy, 0, 0,  0, 0, 1, 0,  0, 0, 0, 1 };
	set_rows( pR, r );
}

rt_float rt_clamp_float( rt_float f, rt_float b, rt_float e ) { return ( f > b ) ? ( ( f < e ) ? f : e ) : b; }
rt_float minF( rt_float a, rt_float b ) { return ( a < b ) ? a : b; }
rt_float maxF( rt_float a, rt_float b ) { return ( a > b ) ? a : b; }

rt_float rt_vector3_length_quad( rt_vector3 *pV ) { return pV->x * pV->x + pV->y * pV->y + pV->z * pV->z; }
rt_float rt_vector3_length( rt_vector3 *pV ) { return (float) sqrt( rt_vector3_length_quad( pV ) ); }

void rt_vector3_normalize( rt_vector3 *pV, rt_vector3 *pR )
{
	const float len = rt_vector3_length( pV );
	const float x = pV->x / len, y = pV->y / len, z = pV->z / len;
	pR->x = x; pR->y = y; pR->z = z;
}

rt_float rt_vector3_distance_quad( rt_vector3 *pV0, rt_vector3 *pV1 )
{
	const float dx = pV0->x - pV1->x, dy = pV0->y - pV1->y, dz = pV0->z - pV1->z;
	return dx * dx + dy * dy + dz * dz;
}
rt_float rt_vector3_distance( rt_vector3 *pV0, rt_vector3 *pV1 ) { return (float) sqrt( rt_vector3_distance_quad( pV0, pV1 ) ); }

void rt_vector3_scalar_mult( rt_vector3 *pV, rt_float s, rt_vector3 *pR )
{
	pR->x = pV->x * s; pR->y = pV->y * s; pR->z = pV->z * s;
}

// NOTE the operand order: second minus first
void rt_vector3_sub( rt_vector3 *pV0, rt_vector3 *pV1, rt_vector3 *pR )
{
	pR->x = pV1->x - pV0->x; pR->y = pV1->y - pV0->y; pR->z = pV1->z - pV0->z;
}

void rt_vector3_add( rt_vector3 *pV0, rt_vector3 *pV1, rt_vector3 *pR )
{
	pR->x = pV1->x + pV0->x; pR->y = pV1->y + pV0->y; pR->z = pV1->z + pV0->z;
}

rt_float rt_vector3_dot( rt_vector3 *pV0, rt_vector3 *pV1 ) { return pV0->x * pV1->x + pV0->y * pV1->y + pV0->z * pV1->z; }

void rt_vector3_cross( rt_vector3 *pV0, rt_vector3 *pV1, rt_vector3 *pR )
{
	const float x = pV0->y * pV1->z - pV0->z * pV1->y;
	const float y = pV0->z * pV1->x - pV0->x * pV1->z;
	const float z = pV0->x * pV1->y - pV0->y * pV1->x;
	pR->x = x; pR->y = y; pR->z = z;
}

// v - 2 (v.n) n, built from the swapped-operand subtraction like the reference
void rt_vector3_reflect( rt_vector3 *pV, rt_vector3 *pN, rt_vector3 *pR )
{
	rt_vector3 t;
	rt_vector3_scalar_mult( pN, 2.0f * rt_vector3_dot( pV, pN ), &t );
	rt_vector3_sub( &t, pV, pR );
}

void rt_vector3_matrix3_mult( rt_vector3 *pV, rt_matrix3 *pM, rt_vector3 *pR ) { vec_mat( pV, pM, false, pR ); }
void rt_vector3_matrix4_mult( rt_vector3 *pV, rt_matrix4 *pM, rt_vector3 *pR ) { vec_mat( pV, pM, true, pR ); }
void rt_vector3_matrix4_mult_dir( rt_vector3 *pV, rt_matrix4 *pM, rt_vector3 *pR ) { vec_mat( pV, pM, false, pR ); }

void rt_color_clamp( rt_color *pC, rt_color *pR )
{
	pR->a = rt_clamp_float( pC->a, 0.0f, 1.0f );
	pR->r = rt_clamp_float( pC->r, 0.0f, 1.0f );
	pR->g = rt_clamp_float( pC->g, 0.0f, 1.0f );
	pR->b = rt_clamp_float( pC->b, 0.0f, 1.0f );
}

// host colour ops include the alpha channel (the device ones do not)
void rt_color_mult( rt_color *pC0, rt_color *pC1, rt_color *pR )
{
	rt_color t = { pC0->b * pC1->b, pC0->g * pC1->g, pC0->r * pC1->r, pC0->a * pC1->a };
	rt_color_clamp( &t, pR );
}
void rt_color_add( rt_color *pC0, rt_color *pC1, rt_color *pR )
{
	rt_color t = { pC0->b + pC1->b, pC0->g + pC1->g, pC0->r + pC1->r, pC0->a + pC1->a };
	rt_color_clamp( &t, pR );
}
void rt_color_scalar_mult( rt_color *pC, rt_float s, rt_color *pR )
{
	rt_color t = { pC->b * s, pC->g * s, pC->r * s, pC->a * s };
	rt_color_clamp( &t, pR );
}

void rt_matrix4_inverse( rt_matrix4 *pM, rt_matrix4 *pR ) { gauss_jordan( M( pM ), 4, M( pR ) ); }

// The reference views the first nine floats of its rt_matrix3 as a packed
// 3x3 array here (rt_funcs_math.c:364-398) and copies sizeof(rt_matrix4)
// bytes of a 36-byte local back; only the nine defined floats are written.
void rt_matrix3_inverse( rt_matrix3 *pM, rt_matrix3 *pR ) { gauss_jordan( M( pM ), 3, M( pR ) ); }

// swaps the mirrored pairs; diagonal entries other than _44 are left as they are
void rt_matrix4_transpose( rt_matrix4 *pM, rt_matrix4 *pR )
{
	float *a = M( pM ), *r = M( pR );
	static const int pairs[6][2] = { { 1, 4 }, { 2, 8 }, { 3, 12 }, { 6, 9 }, { 7, 13 }, { 11, 14 } };
	for ( auto &p : pairs )
	{
		const float t = a[p[0]];
		r[p[0]] = a[p[1]];
		r[p[1]] = t;
	}
	r[15] = a[15];
}

void rt_matrix3_transpose( rt_matrix3 *pM, rt_matrix3 *pR )
{
	float *a = M( pM ), *r = M( pR );
	static const int pairs[3][2] = { { 1, 4 }, { 2, 8 }, { 6, 9 } };
	for ( auto &p : pairs )
	{
		const float t = a[p[0]];
		r[p[0]] = a[p[1]];
		r[p[1]] = t;
	}
}

void rt_matrix4_mult( rt_matrix4 *pM0, rt_matrix4 *pM1, rt_matrix4 *pR ) { mat_mult4( pM0, pM1, pR ); }

// 3x3 product on the members _11.._33; the reference copies its result with
// memcpy( pR, &tmp, sizeof(rt_matrix3) ) where rt_matrix3 is the 4x4 struct,
// so all sixteen floats of pR are overwritten: the 3x3 block with the
// product, the rest with the (indeterminate) remainder of its local.  Here
// the remainder keeps pR's previous contents.
void rt_matrix3_mult( rt_matrix3 *pM0, rt_matrix3 *pM1, rt_matrix3 *pR )
{
	const float *a = M( pM0 ), *b = M( pM1 );
	float r[16];
	memcpy( r, M( pR ), 64 );
	float p[9];
	for ( int i = 0; i < 3; ++i )
		for ( int j = 0; j < 3; ++j )
			p[3 * i + j] = ( b[j] * a[4 * i] + b[4 + j] * a[4 * i + 1] ) + b[8 + j] * a[4 * i + 2];
	for ( int i = 0; i < 3; ++i )
		for ( int j = 0; j < 3; ++j )
			r[4 * i + j] = p[3 * i + j];
	memcpy( M( pR ), r, 64 );
}

int is_power_of_two_size_t( size_t a )
{
	// at most one bit set (zero counts, as in the reference)
	return ( a & ( a - 1 ) ) == 0;
}

} // extern "C"
