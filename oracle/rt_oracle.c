/*
 * CPU oracle: TEST INFRASTRUCTURE ONLY (see rt_oracle.h for the parity
 * status: tree builder pinned, device kernel "parity unpinned").
 *
 * Plain scalar fp32 C, built with -ffp-contract=off so every product and
 * sum rounds once, in source order.  Each function names the reference
 * lines it restates.  OpenCL built-ins are restated as:
 *   dot(a,b)      = (a.x*b.x + a.y*b.y) + a.z*b.z
 *   cross(a,b)    = (a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x)
 *   length(v)     = sqrtf(dot(v,v))
 *   normalize(v)  = v / length(v)          (three divisions)
 *   max(x,y)      = (x < y) ? y : x
 *   sqrt          = sqrtf
 *   pow, exp      = the double-precision function rounded to float, i.e. the
 *                   correctly rounded single-precision value in all but about
 *                   one case in 2^29 (OpenCL leaves the rounding of these two
 *                   to the vendor; the HIP kernels compute them the same way,
 *                   so both sides agree to the bit)
 *
 * Behaviour the reference leaves undefined is resolved as follows (the HIP
 * kernels resolve it the same way):
 *   - a primitive descriptor that is not a sphere never intersects
 *     (the reference reads an uninitialised flag, CL/rt_funcs_primitives.cl:390-404);
 *   - a closest-hit query that reports "hit" without a valid distance (flag
 *     set in an earlier leaf, distance reset by a later one, :82 vs :106-107)
 *     and no sphere hit is shaded as a miss;
 *   - the hit point of a ray that misses is (0,0,0) (:604, read at :708).
 */
#include "rt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

static inline float cl_pow( float x, float y ) { return (float) pow( (double) x, (double) y ); }
static inline float cl_exp( float x ) { return (float) exp( (double) x ); }

typedef struct { float x, y, z; } v3;
typedef struct { v3 src, dest, inv; } ray_t;
typedef struct { float r, g, b; } rgb_t;

static inline v3 V( float x, float y, float z ) { v3 r = { x, y, z }; return r; }
static inline v3 ld3( const rt_vector3 *p ) { return V( p->x, p->y, p->z ); }
static inline v3 add3( v3 a, v3 b ) { return V( a.x + b.x, a.y + b.y, a.z + b.z ); }
static inline v3 sub3( v3 a, v3 b ) { return V( a.x - b.x, a.y - b.y, a.z - b.z ); }
static inline v3 mul3( v3 a, float s ) { return V( a.x * s, a.y * s, a.z * s ); }
static inline v3 div3( v3 a, float s ) { return V( a.x / s, a.y / s, a.z / s ); }
static inline v3 rcp3( v3 a ) { return V( 1.0f / a.x, 1.0f / a.y, 1.0f / a.z ); }
static inline float dot3( v3 a, v3 b ) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 cross3( v3 a, v3 b )
{
	return V( a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x );
}
static inline float len3( v3 a ) { return sqrtf( dot3( a, a ) ); }
static inline v3 norm3( v3 a ) { return div3( a, len3( a ) ); }
static inline float cl_max( float x, float y ) { return ( x < y ) ? y : x; }

/* CL/rt_funcs_math.cl:102-105 */
static inline float clampf( float f, float b, float e )
{
	return ( f > b ) ? ( ( f < e ) ? f : e ) : b;
}
static inline rgb_t clamp_rgb( rgb_t c )
{
	rgb_t r = { clampf( c.r, 0.0f, 1.0f ), clampf( c.g, 0.0f, 1.0f ), clampf( c.b, 0.0f, 1.0f ) };
	return r;
}
/* CL/rt_funcs_math.cl:157-185: every colour op clamps r,g,b */
static inline rgb_t col_mult( rgb_t a, rgb_t b )
{
	rgb_t r = { a.r * b.r, a.g * b.g, a.b * b.b };
	return clamp_rgb( r );
}
static inline rgb_t col_add( rgb_t a, rgb_t b )
{
	rgb_t r = { a.r + b.r, a.g + b.g, a.b + b.b };
	return clamp_rgb( r );
}
static inline rgb_t col_scale( rgb_t a, float s )
{
	rgb_t r = { a.r * s, a.g * s, a.b * s };
	return clamp_rgb( r );
}
static inline rgb_t ld_rgb( const rt_color *c ) { rgb_t r = { c->r, c->g, c->b }; return r; }

/* CL/rt_funcs_math.cl:119-124: pV - pN * 2.0f * dot(pV,pN) */
static inline v3 reflect3( v3 v, v3 n )
{
	float d = dot3( v, n );
	return sub3( v, mul3( mul3( n, 2.0f ), d ) );
}
/* CL/rt_funcs_math.cl:126-147 */
static inline v3 mat4_point( v3 v, const rt_matrix4 *m )
{
	return V( m->_11 * v.x + m->_21 * v.y + m->_31 * v.z + m->_41,
	          m->_12 * v.x + m->_22 * v.y + m->_32 * v.z + m->_42,
	          m->_13 * v.x + m->_23 * v.y + m->_33 * v.z + m->_43 );
}
static inline v3 mat4_dir( v3 v, const rt_matrix4 *m )
{
	return V( m->_11 * v.x + m->_21 * v.y + m->_31 * v.z,
	          m->_12 * v.x + m->_22 * v.y + m->_32 * v.z,
	          m->_13 * v.x + m->_23 * v.y + m->_33 * v.z );
}
static inline float minF( float a, float b ) { return ( a < b ) ? a : b; }   /* :300-308 */
static inline float maxF( float a, float b ) { return ( a > b ) ? a : b; }

/* ------------------------------------------------------------------------ *
 * intersections, CL/rt_intersection.cl
 * ------------------------------------------------------------------------ */

/* :7-64 */
static int ray_sphere( const ray_t *r, v3 c, float rad, float *t )
{
	v3 tmp = sub3( r->src, c );
	float b = dot3( r->dest, tmp );
	float cc, d, x1, x2;

	b *= 2.0f;
	cc = dot3( tmp, tmp );
	cc -= rad * rad;
	d = b * b - 4 * cc;
	if ( d < 0.0f )
		return 0;
	d = sqrtf( d );
	x1 = ( -b - d ) / 2.0f;
	x2 = ( -b + d ) / 2.0f;
	if ( x2 < 0.0f )
		return 0;
	if ( x1 < 0.0f )
	{
		*t = x2;
		return -1;
	}
	*t = x1;
	return 1;
}

/* :110-154, Moeller-Trumbore, two sided, exact det == 0 rejection */
static int ray_triangle( const ray_t *r, v3 v0, v3 v1, v3 v2, float *t, float *u, float *v )
{
	v3 e1 = sub3( v1, v0 );
	v3 e2 = sub3( v2, v0 );
	v3 p = cross3( r->dest, e2 );
	float det = dot3( e1, p );
	v3 dist, q;

	if ( det == 0.0f )
		return 0;
	det = 1.0f / det;
	*t = 0.0f;
	dist = sub3( r->src, v0 );
	*u = dot3( p, dist ) * det;
	if ( ( *u < 0.0f ) || ( *u > 1.0f ) )
		return 0;
	q = cross3( dist, e1 );
	*v = dot3( q, r->dest ) * det;
	if ( ( *v < 0.0f ) || ( *v + *u > 1.0f ) )
		return 0;
	*t = dot3( q, e2 ) * det;
	if ( *t > 0.0f )
		return 1;
	return 0;
}

/* :156-200, slab test with precomputed reciprocal direction */
static int box_ray( const rt_box *c, const ray_t *r, float *tNear, float *tFar )
{
	float t1 = ( c->center.x - c->extents.x - r->src.x ) * r->inv.x;
	float t2 = ( c->center.x + c->extents.x - r->src.x ) * r->inv.x;
	float t3 = ( c->center.y - c->extents.y - r->src.y ) * r->inv.y;
	float t4 = ( c->center.y + c->extents.y - r->src.y ) * r->inv.y;
	float t5 = ( c->center.z - c->extents.z - r->src.z ) * r->inv.z;
	float t6 = ( c->center.z + c->extents.z - r->src.z ) * r->inv.z;

	*tNear = maxF( maxF( minF( t1, t2 ), minF( t3, t4 ) ), minF( t5, t6 ) );
	*tFar = minF( minF( maxF( t1, t2 ), maxF( t3, t4 ) ), maxF( t5, t6 ) );
	if ( *tFar < 0 )
		return 0;
	if ( *tNear > *tFar )
		return 0;
	return 1;
}

/* ------------------------------------------------------------------------ *
 * per-pixel tracer, CL/rt_funcs_primitives.cl
 * ------------------------------------------------------------------------ */

typedef struct
{
	rt_box boundingBox;
	const unsigned char *primsBuf;
	const rt_prim_desc *primsDecs;
	const unsigned char *lightsBuf;
	const rt_light_desc *lightsDecs;
	const rt_material *materialBuf;
	const rt_triangle *trianglesBuf;
	const rt_vertex *vertexBuf;
	const rt_cl_kdtree_node *nodes;
	const uint64_t *primsIdx;
	uint64_t primsCount, lightsCount, trianglesCount;
	int stackSize, kdStackCap;
	int inShadow;              /* 1 while a shadow query runs (statistics only) */
	orc_stats st;
} pipe_t;

static inline int tri_test( pipe_t *p, const ray_t *r, uint64_t tri, float *t, float *u, float *v )
{
	const rt_triangle *tr = p->trianglesBuf + tri;
	++p->st.triTests;
	p->st.triTestsShadow += p->inShadow;
	return ray_triangle( r, ld3( &p->vertexBuf[tr->pV0].pos ), ld3( &p->vertexBuf[tr->pV1].pos ),
		ld3( &p->vertexBuf[tr->pV2].pos ), t, u, v );
}

/* :74-111 - note the per-leaf reset of *minT while *prN / *nearestB persist */
static void nearest_in_last( pipe_t *p, const ray_t *r, uint64_t nodeIdx, float *minT,
	float *minU, float *minV, uint64_t *prN, int *nearestB )
{
	uint64_t i, sz = p->nodes[nodeIdx].primsCount;

	*minT = INFINITY;
	for ( i = 0; i < sz; ++i )
	{
		float t, u, v;
		uint64_t tri = p->primsIdx[p->nodes[nodeIdx].prims + i];
		int b = tri_test( p, r, tri, &t, &u, &v );

		if ( b && t < *minT )
		{
			*minT = t;
			*minU = u;
			*minV = v;
			*prN = tri;
			*nearestB = b;
		}
	}
}

/* shared inner-node step of :185-251 and :312-377; returns the next node */
static inline uint64_t kd_step( const rt_cl_kdtree_node *n, const ray_t *r, float tNear,
	float *tFar, uint64_t *stack, float *stackFar, int *stackPos, int cap )
{
	float sep = n->sep, tSplit = 0.0f;
	uint64_t nearNode = n->leftNode, farNode = n->rightNode;
	int swap = 0;

	switch ( n->axis )
	{
	case RT_AXIS_X: tSplit = ( sep - r->src.x ) * r->inv.x; swap = r->dest.x < 0.0f; break;
	case RT_AXIS_Y: tSplit = ( sep - r->src.y ) * r->inv.y; swap = r->dest.y < 0.0f; break;
	case RT_AXIS_Z: tSplit = ( sep - r->src.z ) * r->inv.z; swap = r->dest.z < 0.0f; break;
	}
	if ( swap )
	{
		uint64_t tmp = nearNode;
		nearNode = farNode;
		farNode = tmp;
	}
	if ( tSplit >= *tFar )
		return nearNode;
	if ( tSplit <= tNear )
		return farNode;
	if ( *stackPos >= cap )
		abort();        /* tree deeper than the traversal stack: caller error */
	stack[*stackPos] = farNode;
	stackFar[*stackPos] = *tFar;
	++*stackPos;
	*tFar = tSplit;
	return nearNode;
}

#define ORC_KD_STACK_MAX 128

/* :151-253 */
static void nearest_triangle( pipe_t *p, const ray_t *r, float *minT, float *minU, float *minV,
	uint64_t *prN, int *nearestB )
{
	float tNear, tFar;
	uint64_t cur = 0;
	uint64_t stack[ORC_KD_STACK_MAX];
	float stackFar[ORC_KD_STACK_MAX];
	int stackPos = 0;

	++p->st.raysClosest;
	*minT = INFINITY;
	if ( !box_ray( &p->boundingBox, r, &tNear, &tFar ) )
		return;
	for ( ;; )
	{
		const rt_cl_kdtree_node *n = p->nodes + cur;
		++p->st.nodeVisits;
		if ( n->isLast )
		{
			nearest_in_last( p, r, cur, minT, minU, minV, prN, nearestB );
			if ( *nearestB && ( *minT < tFar ) )
				return;
			if ( !stackPos )
				return;
			--stackPos;
			cur = stack[stackPos];
			tNear = tFar;
			tFar = stackFar[stackPos];
		}
		else
			cur = kd_step( n, r, tNear, &tFar, stack, stackFar, &stackPos, p->kdStackCap );
	}
}

/* :256-276 */
static void add_alpha_in_last( pipe_t *p, const ray_t *r, uint64_t nodeIdx, float d,
	float tNear, float tFar, float *alpha )
{
	uint64_t i, sz = p->nodes[nodeIdx].primsCount;

	for ( i = 0; i < sz; ++i )
	{
		float t, u, v;
		uint64_t tri = p->primsIdx[p->nodes[nodeIdx].prims + i];
		int b = tri_test( p, r, tri, &t, &u, &v );

		if ( b && ( t < d ) && ( t < tFar ) && ( t > tNear ) )
			*alpha += p->materialBuf[p->trianglesBuf[tri].mat].color.a;
	}
}

/* :278-379 - *alpha is written only once the scene box is hit */
static void alpha_triangles( pipe_t *p, const ray_t *r, float d, float *alpha )
{
	float tNear, tFar;
	uint64_t cur = 0;
	uint64_t stack[ORC_KD_STACK_MAX];
	float stackFar[ORC_KD_STACK_MAX];
	int stackPos = 0;

	++p->st.raysShadow;
	if ( !box_ray( &p->boundingBox, r, &tNear, &tFar ) )
		return;
	*alpha = 0.0f;
	p->inShadow = 1;
	for ( ;; )
	{
		const rt_cl_kdtree_node *n = p->nodes + cur;
		++p->st.nodeVisits;
		++p->st.nodeVisitsShadow;
		if ( n->isLast )
		{
			add_alpha_in_last( p, r, cur, d, tNear, tFar, alpha );
			if ( !stackPos )
			{
				p->inShadow = 0;
				return;
			}
			--stackPos;
			cur = stack[stackPos];
			tNear = tFar;
			tFar = stackFar[stackPos];
		}
		else
			cur = kd_step( n, r, tNear, &tFar, stack, stackFar, &stackPos, p->kdStackCap );
	}
}

/* :381-416 */
static void nearest_prim( pipe_t *p, const ray_t *r, float *minT, uint64_t *prN, int *nearestB )
{
	uint64_t i;

	*minT = INFINITY;
	for ( i = 0; i < p->primsCount; ++i )
	{
		int b = 0;
		float t = 0.0f;

		if ( p->primsDecs[i].pt == RT_PT_SPHERE )
		{
			const rt_sphere *s = (const rt_sphere *) ( p->primsBuf + p->primsDecs[i].offset );
			b = ray_sphere( r, ld3( &s->pos ), s->rad, &t );
		}
		if ( b && t < *minT )
		{
			*minT = t;
			*prN = i;
			*nearestB = b;
		}
	}
}

/* :418-448 */
static void alpha_prims( pipe_t *p, const ray_t *r, float d, float *alpha )
{
	uint64_t i;

	*alpha = 0.0f;
	for ( i = 0; i < p->primsCount; ++i )
	{
		int b = 0;
		float t = 0.0f;

		if ( p->primsDecs[i].pt == RT_PT_SPHERE )
		{
			const rt_sphere *s = (const rt_sphere *) ( p->primsBuf + p->primsDecs[i].offset );
			b = ray_sphere( r, ld3( &s->pos ), s->rad, &t );
			if ( b && t < d )
				*alpha += p->materialBuf[s->mat].color.a;
		}
	}
}

/* :450-564 */
static rgb_t light_point( pipe_t *p, v3 P, v3 N, const rt_material *mat, v3 viewer )
{
	rgb_t col = { 0.0f, 0.0f, 0.0f };
	uint64_t o;

	for ( o = 0; o < p->lightsCount; ++o )
	{
		const rt_point_light *L;
		rgb_t lcol, ambient, diffuse, specular, tmp;
		v3 toLight, toEye;
		float d, ang, shadowed, a = 0.0f, specFact;
		ray_t sh;

		if ( p->lightsDecs[o].lt != RT_LT_POINT )
			continue;
		L = (const rt_point_light *) ( p->lightsBuf + p->lightsDecs[o].offset );
		lcol = ld_rgb( &L->col );

		toLight = sub3( ld3( &L->pos ), P );
		d = len3( toLight );
		toLight = div3( toLight, d );

		/* shadows :481-504 (alpha accumulates twice when the shadow ray
		 * misses the scene box, because `a` keeps the sphere sum) */
		sh.dest = toLight;
		sh.src = add3( P, mul3( toLight, EPSILON ) );
		sh.inv = rcp3( toLight );
		alpha_prims( p, &sh, d, &a );
		shadowed = clampf( a, 0.0f, 1.0f );
		alpha_triangles( p, &sh, d, &a );
		shadowed += clampf( a, 0.0f, 1.0f );
		shadowed *= mat->color.a;

		ambient = col_mult( lcol, ld_rgb( &mat->ambient ) );                       /* :506-511 */

		ang = cl_max( 0.0f, dot3( N, toLight ) );                                  /* :514-530 */
		diffuse = col_mult( lcol, ld_rgb( &mat->diffuse ) );
		diffuse = col_mult( diffuse, ld_rgb( &mat->color ) );
		diffuse = col_scale( diffuse, ang * L->rad / d );

		toLight = mul3( toLight, -1.0f );                                          /* :532-556 */
		toLight = reflect3( toLight, N );
		toEye = norm3( sub3( viewer, P ) );
		ang = dot3( toEye, toLight );
		specFact = cl_pow( cl_max( ang, 0.0f ), mat->specular.a );
		specular = col_mult( ld_rgb( &mat->specular ), lcol );
		specular = col_scale( specular, specFact * L->rad / d );

		diffuse = col_add( ambient, diffuse );                                     /* :558-562 */
		tmp = col_add( diffuse, specular );
		tmp = col_scale( tmp, 1.0f - shadowed );
		col = col_add( tmp, col );
	}
	return col;
}

typedef struct
{
	ray_t ray;
	rgb_t col;
	v3 p;
	uint64_t mat;
	int parentIdx, bNearest, type;
} selem_t;

/* :584-728; *primaryHit receives the hit id of the first ray */
static rgb_t raytrace_px( pipe_t *p, const ray_t *pR, int32_t *primaryHit )
{
	selem_t stack[RT_STACK_SIZE_MAX];
	int stackPos = 0, curNode = 0;
	const int cap = p->stackSize;
	rt_material tmpMat;

	stack[0].parentIdx = -1;
	stack[0].ray = *pR;

	for ( ;; )
	{
		float prMinT, trMinT, trMinU = 0.0f, trMinV = 0.0f;
		uint64_t prN = 0, trN = 0;
		int prNearestB = 0, trNearestB = 0, hit = 0;
		v3 P = V( 0.0f, 0.0f, 0.0f ), N = V( 0.0f, 0.0f, 0.0f );
		selem_t *e = stack + curNode;

		e->col.r = e->col.g = e->col.b = 0.0f;     /* fillCol, CL/rt_raytrace.cl:34-37 */
		e->p = P;
		e->mat = 0;

		nearest_prim( p, &e->ray, &prMinT, &prN, &prNearestB );
		nearest_triangle( p, &e->ray, &trMinT, &trMinU, &trMinV, &trN, &trNearestB );

		if ( trNearestB || prNearestB )
		{
			if ( trMinT < prMinT )
			{
				/* :613-624 + :62-72, normal neither normalised nor flipped */
				const rt_triangle *tr = p->trianglesBuf + trN;
				v3 n0 = ld3( &p->vertexBuf[tr->pV0].norm );
				v3 n1 = ld3( &p->vertexBuf[tr->pV1].norm );
				v3 n2 = ld3( &p->vertexBuf[tr->pV2].norm );

				e->mat = tr->mat;
				e->bNearest = trNearestB;
				P = add3( mul3( e->ray.dest, trMinT ), e->ray.src );
				N = add3( add3( mul3( n1, trMinU ), mul3( n2, trMinV ) ),
					mul3( n0, 1.0f - trMinU - trMinV ) );
				hit = 1;
				if ( curNode == 0 && primaryHit )
					*primaryHit = (int32_t) trN;
			}
			else if ( prNearestB )
			{
				/* :625-633 + :31-49 */
				const rt_sphere *s = (const rt_sphere *) ( p->primsBuf + p->primsDecs[prN].offset );

				e->mat = s->mat;
				e->bNearest = prNearestB;
				P = mul3( e->ray.dest, prMinT );
				P = add3( P, e->ray.src );
				N = div3( sub3( P, ld3( &s->pos ) ), s->rad );
				N = mul3( N, (float) prNearestB );
				hit = 1;
				if ( curNode == 0 && primaryHit )
					*primaryHit = ORC_HIT_SPHERE( prN );
			}
			else
				++p->st.staleHits;
		}

		if ( hit )
		{
			++p->st.hits;
			e->p = P;
			tmpMat = p->materialBuf[e->mat];
			e->col = light_point( p, P, N, &tmpMat, e->ray.src );

			/* reflection ray :643-655 */
			if ( ( ( stackPos + 1 ) < cap ) && ( e->bNearest != -1 )
				&& ( ( tmpMat.reflect.r + tmpMat.reflect.g + tmpMat.reflect.b ) > 0.0f ) )
			{
				selem_t *c = stack + ++stackPos;
				c->ray.dest = reflect3( e->ray.dest, N );
				c->ray.src = add3( P, mul3( c->ray.dest, EPSILON ) );
				c->ray.inv = rcp3( c->ray.dest );
				c->parentIdx = curNode;
				c->type = 1;
			}
			/* refraction ray :657-677 */
			if ( ( ( stackPos + 1 ) < cap ) && ( tmpMat.color.a < 0.99f ) )
			{
				float nRel = ENV_OPT_DENSITY / tmpMat.optDens;
				float cosI = dot3( N, e->ray.dest );
				float sinT = nRel * nRel * ( 1.0f - cosI * cosI );

				if ( sinT < 1.0f )
				{
					selem_t *c = stack + ++stackPos;
					c->ray.dest = norm3( sub3( mul3( e->ray.dest, nRel ),
						mul3( N, nRel * cosI + sqrtf( 1.0f - sinT ) ) ) );
					c->ray.src = add3( P, mul3( c->ray.dest, EPSILON ) );
					c->ray.inv = rcp3( c->ray.dest );
					c->parentIdx = curNode;
					c->type = 0;
				}
			}
		}
		else
			e->bNearest = 0;

		if ( curNode == stackPos )
			break;
		++curNode;
	}

	/* fold children into parents, last to first :688-725 */
	while ( curNode )
	{
		selem_t *c = stack + curNode, *par = stack + c->parentIdx;
		const rt_material *m = p->materialBuf + par->mat;

		if ( c->type == 1 )
		{
			par->col.r += c->col.r * m->reflect.r;
			par->col.g += c->col.g * m->reflect.g;
			par->col.b += c->col.b * m->reflect.b;
		}
		else
		{
			float dist = len3( sub3( par->p, c->p ) );
			float rr = cl_exp( m->color.r * m->lightFalloff * -dist );
			float rg = cl_exp( m->color.g * m->lightFalloff * -dist );
			float rb = cl_exp( m->color.b * m->lightFalloff * -dist );

			par->col.r += c->col.r * rr;
			par->col.g += c->col.g * rg;
			par->col.b += c->col.b * rb;
		}
		par->col = clamp_rgb( par->col );
		--curNode;
	}
	return stack[0].col;
}

/* camera ray of pixel (x,y), CL/rt_raytrace.cl:27-28,60-79 */
static ray_t primary_ray( const rt_camera *cam, int w, int h, int x, int y )
{
	float xTr = 2.0f / (float) ( w - 1 );
	float yTr = 2.0f / (float) ( h - 1 );
	ray_t r;

	r.src = V( 0.0f, 0.0f, 0.0f );
	r.dest.z = 1.0f;
	r.dest.x = xTr * (float) x - 1.0f;
	r.dest.y = yTr * (float) y - 1.0f;
	r.src = mat4_point( r.src, &cam->world );
	r.dest = mat4_dir( r.dest, &cam->viewToPersp );
	r.dest = mat4_dir( r.dest, &cam->world );
	r.dest = norm3( r.dest );
	r.inv = rcp3( r.dest );
	return r;
}

typedef struct
{
	pipe_t pipe;
	const rt_camera *cam;
	int w, h, x0, xStep, y0, yStep, tid, nThreads;
	rt_argb *o;
	int32_t *hitId;
	float *hitT;
	int brute;
} job_t;

static void brute_px( pipe_t *p, const ray_t *r, int32_t *id, float *tOut )
{
	float minT = INFINITY, prMinT;
	uint64_t i, prN = 0;
	int prB = 0;

	*id = ORC_HIT_MISS;
	for ( i = 0; i < p->trianglesCount; ++i )
	{
		float t, u, v;
		if ( tri_test( p, r, i, &t, &u, &v ) && t < minT )
		{
			minT = t;
			*id = (int32_t) i;
		}
	}
	nearest_prim( p, r, &prMinT, &prN, &prB );
	if ( prB && !( minT < prMinT ) )
	{
		*id = ORC_HIT_SPHERE( prN );
		minT = prMinT;
	}
	if ( tOut )
		*tOut = minT;
}

static void *worker( void *arg )
{
	job_t *j = (job_t *) arg;
	int row = 0, x, y;

	for ( y = j->y0; y < j->h; y += j->yStep, ++row )
	{
		if ( row % j->nThreads != j->tid )
			continue;
		for ( x = j->x0; x < j->w; x += j->xStep )
		{
			ray_t r = primary_ray( j->cam, j->w, j->h, x, y );
			size_t px = (size_t) y * j->w + x;

			if ( j->brute )
			{
				int32_t id;
				float t;
				brute_px( &j->pipe, &r, &id, &t );
				if ( j->hitId ) j->hitId[px] = id;
				if ( j->hitT ) j->hitT[px] = t;
			}
			else
			{
				int32_t id = ORC_HIT_MISS;
				rgb_t c = raytrace_px( &j->pipe, &r, &id );
				/* store: truncating float -> uchar, alpha always 0 (CL/rt_raytrace.cl:83-86) */
				j->o[px].a = (unsigned char) ( 255.0f * 0.0f );
				j->o[px].r = (unsigned char) ( 255.0f * c.r );
				j->o[px].g = (unsigned char) ( 255.0f * c.g );
				j->o[px].b = (unsigned char) ( 255.0f * c.b );
				if ( j->hitId ) j->hitId[px] = id;
			}
		}
	}
	return NULL;
}

static void run_jobs( job_t *proto, int nThreads, orc_stats *stats )
{
	pthread_t *th;
	job_t *jobs;
	int i;

	if ( nThreads < 1 ) nThreads = 1;
	if ( nThreads > 256 ) nThreads = 256;
	th = (pthread_t *) malloc( sizeof(pthread_t) * nThreads );
	jobs = (job_t *) malloc( sizeof(job_t) * nThreads );
	for ( i = 0; i < nThreads; ++i )
	{
		jobs[i] = *proto;
		jobs[i].tid = i;
		jobs[i].nThreads = nThreads;
		memset( &jobs[i].pipe.st, 0, sizeof(orc_stats) );
		if ( i > 0 )
			pthread_create( th + i, NULL, worker, jobs + i );
	}
	worker( jobs );
	for ( i = 1; i < nThreads; ++i )
		pthread_join( th[i], NULL );
	if ( stats )
	{
		memset( stats, 0, sizeof(*stats) );
		for ( i = 0; i < nThreads; ++i )
		{
			stats->raysClosest += jobs[i].pipe.st.raysClosest;
			stats->raysShadow += jobs[i].pipe.st.raysShadow;
			stats->nodeVisits += jobs[i].pipe.st.nodeVisits;
			stats->triTests += jobs[i].pipe.st.triTests;
			stats->hits += jobs[i].pipe.st.hits;
			stats->staleHits += jobs[i].pipe.st.staleHits;
			stats->nodeVisitsShadow += jobs[i].pipe.st.nodeVisitsShadow;
			stats->triTestsShadow += jobs[i].pipe.st.triTestsShadow;
		}
	}
	free( th );
	free( jobs );
}

void orc_raytrace( const rt_raytrace_args *a, const void *prims, const rt_prim_desc *pd,
	const void *lights, const rt_light_desc *ld, const rt_camera *cams,
	const rt_material *materials, const rt_triangle *triangles, const rt_vertex *verticies,
	const rt_cl_kdtree_node *kdtreeNodes, const uint64_t *kdtreePrimsIdx, rt_argb *o,
	int stackSize, int kdStackCap, int x0, int xStep, int y0, int yStep,
	int32_t *hitId, orc_stats *stats, int nThreads )
{
	job_t j;

	memset( &j, 0, sizeof(j) );
	if ( stackSize < 1 ) stackSize = 1;
	if ( stackSize > RT_STACK_SIZE_MAX ) stackSize = RT_STACK_SIZE_MAX;
	if ( kdStackCap > ORC_KD_STACK_MAX ) kdStackCap = ORC_KD_STACK_MAX;
	j.pipe.boundingBox = a->boundingBox;
	j.pipe.primsBuf = (const unsigned char *) prims;
	j.pipe.primsDecs = pd;
	j.pipe.lightsBuf = (const unsigned char *) lights;
	j.pipe.lightsDecs = ld;
	j.pipe.materialBuf = materials;
	j.pipe.trianglesBuf = triangles;
	j.pipe.vertexBuf = verticies;
	j.pipe.nodes = kdtreeNodes;
	j.pipe.primsIdx = kdtreePrimsIdx;
	j.pipe.primsCount = a->primsCount;
	j.pipe.lightsCount = a->lightsCount;
	j.pipe.trianglesCount = a->trianglesCount;
	j.pipe.stackSize = stackSize;
	j.pipe.kdStackCap = kdStackCap;
	j.cam = cams;
	j.w = a->w; j.h = a->h;
	j.x0 = x0; j.xStep = xStep < 1 ? 1 : xStep;
	j.y0 = y0; j.yStep = yStep < 1 ? 1 : yStep;
	j.o = o;
	j.hitId = hitId;
	run_jobs( &j, nThreads, stats );
}

void orc_brute_primary_hits( const rt_raytrace_args *a, const void *prims, const rt_prim_desc *pd,
	const rt_camera *cam, const rt_triangle *triangles, const rt_vertex *verticies,
	int x0, int xStep, int y0, int yStep, int32_t *hitId, float *hitT, int nThreads )
{
	job_t j;

	memset( &j, 0, sizeof(j) );
	j.pipe.primsBuf = (const unsigned char *) prims;
	j.pipe.primsDecs = pd;
	j.pipe.trianglesBuf = triangles;
	j.pipe.vertexBuf = verticies;
	j.pipe.primsCount = a->primsCount;
	j.pipe.trianglesCount = a->trianglesCount;
	j.cam = cam;
	j.w = a->w; j.h = a->h;
	j.x0 = x0; j.xStep = xStep < 1 ? 1 : xStep;
	j.y0 = y0; j.yStep = yStep < 1 ? 1 : yStep;
	j.hitId = hitId;
	j.hitT = hitT;
	j.brute = 1;
	run_jobs( &j, nThreads, NULL );
}

static ray_t mk_ray( const float s[3], const float d[3] )
{
	ray_t r;
	r.src = V( s[0], s[1], s[2] );
	r.dest = V( d[0], d[1], d[2] );
	r.inv = rcp3( r.dest );
	return r;
}
int orc_ray_triangle( const float src[3], const float dest[3], const float v0[3],
	const float v1[3], const float v2[3], float *t, float *u, float *v )
{
	ray_t r = mk_ray( src, dest );
	return ray_triangle( &r, V( v0[0], v0[1], v0[2] ), V( v1[0], v1[1], v1[2] ),
		V( v2[0], v2[1], v2[2] ), t, u, v );
}
int orc_ray_sphere( const float src[3], const float dest[3], const float c[3], float rad, float *t )
{
	ray_t r = mk_ray( src, dest );
	return ray_sphere( &r, V( c[0], c[1], c[2] ), rad, t );
}
int orc_ray_box( const rt_box *box, const float src[3], const float dest[3], float *tNear, float *tFar )
{
	ray_t r = mk_ray( src, dest );
	return box_ray( box, &r, tNear, tFar );
}

/* ------------------------------------------------------------------------ *
 * kd-tree builder, reference rt_funcs_render_pipe.c:591-1209
 * ------------------------------------------------------------------------ */

typedef struct
{
	const float *tmin[3], *tmax[3];   /* per-triangle extent on each axis */
	int maxDepth, maxPrims;
	rt_cl_kdtree_node *nodes;
	uint64_t nNodes, capNodes;
	uint64_t *idx;
	uint64_t nIdx, capIdx;
} kdb_t;

static uint64_t kdb_new_node( kdb_t *b )
{
	if ( b->nNodes == b->capNodes )
	{
		b->capNodes = b->capNodes ? b->capNodes * 2 : 1024;
		b->nodes = (rt_cl_kdtree_node *) realloc( b->nodes, b->capNodes * sizeof(rt_cl_kdtree_node) );
	}
	memset( b->nodes + b->nNodes, 0, sizeof(rt_cl_kdtree_node) );
	return b->nNodes++;
}

static inline float box_c( const rt_box *b, int ax ) { return ax == 0 ? b->center.x : ax == 1 ? b->center.y : b->center.z; }
static inline float box_e( const rt_box *b, int ax ) { return ax == 0 ? b->extents.x : ax == 1 ? b->extents.y : b->extents.z; }
static inline void box_set( rt_box *b, int ax, float c, float e )
{
	if ( ax == 0 ) { b->center.x = c; b->extents.x = e; }
	else if ( ax == 1 ) { b->center.y = c; b->extents.y = e; }
	else { b->center.z = c; b->extents.z = e; }
}

/*
 * :591-857.  32 candidate planes sep = min + i*delta; the eight helper
 * threads of the reference each keep their first strict minimum over four
 * consecutive candidates and the caller keeps the first strict minimum over
 * the threads, which is the lowest-index argmin restated here.
 */
static void compute_sah( const kdb_t *b, const uint64_t *prims, uint64_t n, int ax,
	const rt_box *box, uint64_t *cntL, uint64_t *cntR, float *pSep )
{
	float mn = box_c( box, ax ) - box_e( box, ax );
	float mx = box_c( box, ax ) + box_e( box, ax );
	float delta = ( mx - mn ) / (float) SAH_PARTS;
	int o1 = ( ax + 1 ) % 3, o2 = ( ax + 2 ) % 3;
	/* area term and perimeter term in the reference's operand order */
	float eA, eB, area;
	float best = 0.0f;
	int i, first = 1;
	const float *tmin = b->tmin[ax], *tmax = b->tmax[ax];

	if ( ax == 0 ) { eA = box->extents.y; eB = box->extents.z; }        /* yz, (y + z) */
	else if ( ax == 1 ) { eA = box->extents.x; eB = box->extents.z; }   /* xz, (x + z) */
	else { eA = box->extents.x; eB = box->extents.y; }                  /* xy, (x + y) */
	(void) o1; (void) o2;
	area = eA * eB;

	for ( i = 0; i < SAH_PARTS; ++i )
	{
		float sep = mn + i * delta;
		float sah;
		uint64_t j, l = 0, r = 0;

		for ( j = 0; j < n; ++j )
		{
			if ( tmin[prims[j]] < sep ) ++l;
			if ( tmax[prims[j]] > sep ) ++r;
		}
		sah = ( area + ( sep - box_c( box, ax ) + box_e( box, ax ) ) * ( eA + eB ) ) * l
			+ ( area + ( box_c( box, ax ) + box_e( box, ax ) - sep ) * ( eA + eB ) ) * r;
		if ( first || sah < best )
		{
			best = sah;
			*pSep = sep;
			*cntL = l;
			*cntR = r;
			first = 0;
		}
	}
}

/* :914-1089 fused with the pre-order packing of :860-912; returns the node index */
static uint64_t make_childs( kdb_t *b, uint64_t *prims, uint64_t n, const rt_box *box, int depth )
{
	uint64_t self = kdb_new_node( b );
	uint64_t i, j = 0, k = 0, cl = 0, cr = 0, *pl, *pr, right;
	rt_box boxL, boxR;
	float range = box->extents.x, sep = 0.0f;
	int ax = RT_AXIS_X;

	if ( n <= (uint64_t) b->maxPrims || depth == b->maxDepth )
	{
		while ( b->nIdx + n > b->capIdx )
		{
			b->capIdx = b->capIdx ? b->capIdx * 2 : 4096;
			b->idx = (uint64_t *) realloc( b->idx, b->capIdx * sizeof(uint64_t) );
		}
		b->nodes[self].isLast = 1;
		b->nodes[self].primsCount = n;
		b->nodes[self].prims = b->nIdx;
		if ( n )
			memcpy( b->idx + b->nIdx, prims, n * sizeof(uint64_t) );
		b->nIdx += n;
		free( prims );
		return self;
	}

	if ( range < box->extents.y ) { range = box->extents.y; ax = RT_AXIS_Y; }
	if ( range < box->extents.z ) { range = box->extents.z; ax = RT_AXIS_Z; }

	compute_sah( b, prims, n, ax, box, &cl, &cr, &sep );

	pl = (uint64_t *) malloc( sizeof(uint64_t) * ( cl ? cl : 1 ) );
	pr = (uint64_t *) malloc( sizeof(uint64_t) * ( cr ? cr : 1 ) );
	for ( i = 0; i < n; ++i )
	{
		if ( b->tmin[ax][prims[i]] < sep ) pl[j++] = prims[i];
		if ( b->tmax[ax][prims[i]] > sep ) pr[k++] = prims[i];
	}
	free( prims );

	boxL = boxR = *box;
	box_set( &boxL, ax, 0.5f * ( box_c( box, ax ) - box_e( box, ax ) + sep ),
		0.5f * ( sep - box_c( box, ax ) + box_e( box, ax ) ) );
	box_set( &boxR, ax, 0.5f * ( box_c( box, ax ) + box_e( box, ax ) + sep ),
		0.5f * ( box_c( box, ax ) + box_e( box, ax ) - sep ) );

	make_childs( b, pl, cl, &boxL, depth + 1 );
	right = make_childs( b, pr, cr, &boxR, depth + 1 );

	b->nodes[self].sep = sep;
	b->nodes[self].axis = (RT_AXIS) ax;
	b->nodes[self].leftNode = self + 1;
	b->nodes[self].rightNode = right;
	return self;
}

int orc_kdtree_build( const rt_vertex *verts, const rt_triangle *tris, uint64_t vertexCount,
	uint64_t trianglesCount, int maxDepth, int maxPrims, rt_box *bbox,
	rt_cl_kdtree_node **nodes, uint64_t *nodesCount, uint64_t **primsIdx, uint64_t *primsIdxCount )
{
	kdb_t b;
	float *ext[6];
	uint64_t i, *root;
	float mnx = 0, mny = 0, mnz = 0, mxx = 0, mxy = 0, mxz = 0;
	int ax;

	memset( &b, 0, sizeof(b) );
	b.maxDepth = maxDepth;
	b.maxPrims = maxPrims;

	/* scene box :1114-1137: min/max over ALL vertices, +-1, centre / half size */
	for ( i = 0; i < vertexCount; ++i )
	{
		const rt_vector3 *p = &verts[i].pos;
		mnx = ( mnx > p->x || i == 0 ) ? p->x : mnx;
		mny = ( mny > p->y || i == 0 ) ? p->y : mny;
		mnz = ( mnz > p->z || i == 0 ) ? p->z : mnz;
		mxx = ( mxx < p->x || i == 0 ) ? p->x : mxx;
		mxy = ( mxy < p->y || i == 0 ) ? p->y : mxy;
		mxz = ( mxz < p->z || i == 0 ) ? p->z : mxz;
	}
	mnx -= 1.0f; mny -= 1.0f; mnz -= 1.0f;
	mxx += 1.0f; mxy += 1.0f; mxz += 1.0f;
	memset( bbox, 0, sizeof(*bbox) );
	bbox->center.x = ( mnx + mxx ) * 0.5f;
	bbox->center.y = ( mny + mxy ) * 0.5f;
	bbox->center.z = ( mnz + mxz ) * 0.5f;
	bbox->extents.x = ( mxx - mnx ) * 0.5f;
	bbox->extents.y = ( mxy - mny ) * 0.5f;
	bbox->extents.z = ( mxz - mnz ) * 0.5f;

	/* per-triangle extents, the same compare chain as :620-633 / :981-1019 */
	for ( ax = 0; ax < 6; ++ax )
		ext[ax] = (float *) malloc( sizeof(float) * ( trianglesCount ? trianglesCount : 1 ) );
	for ( i = 0; i < trianglesCount; ++i )
	{
		const rt_vector3 *p0 = &verts[tris[i].pV0].pos, *p1 = &verts[tris[i].pV1].pos,
			*p2 = &verts[tris[i].pV2].pos;
		const float c[3][3] = { { p0->x, p1->x, p2->x }, { p0->y, p1->y, p2->y }, { p0->z, p1->z, p2->z } };
		for ( ax = 0; ax < 3; ++ax )
		{
			float mx = c[ax][0], mn = c[ax][0];
			if ( mx < c[ax][1] ) mx = c[ax][1];
			if ( mx < c[ax][2] ) mx = c[ax][2];
			if ( mn > c[ax][1] ) mn = c[ax][1];
			if ( mn > c[ax][2] ) mn = c[ax][2];
			ext[ax][i] = mn;
			ext[3 + ax][i] = mx;
		}
	}
	for ( ax = 0; ax < 3; ++ax )
	{
		b.tmin[ax] = ext[ax];
		b.tmax[ax] = ext[3 + ax];
	}

	root = (uint64_t *) malloc( sizeof(uint64_t) * ( trianglesCount ? trianglesCount : 1 ) );
	for ( i = 0; i < trianglesCount; ++i )
		root[i] = i;

	/* :1146-1164: fewer than MAX_PRIMS_IN_NODE triangles -> one leaf, else recurse
	 * (make_childs turns exactly MAX_PRIMS_IN_NODE into a leaf as well) */
	make_childs( &b, root, trianglesCount, bbox, 0 );

	for ( ax = 0; ax < 6; ++ax )
		free( ext[ax] );
	*nodes = b.nodes;
	*nodesCount = b.nNodes;
	if ( !b.idx )
		b.idx = (uint64_t *) malloc( sizeof(uint64_t) );
	*primsIdx = b.idx;
	*primsIdxCount = b.nIdx;
	return 0;
}

void orc_free( void *p ) { free( p ); }
