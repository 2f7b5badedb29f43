// Mesh container + Wavefront OBJ / MTL loader behind include/rt_meshes.h
// (reference rt_meshes.c).  Same file grammar and the same resulting
// vertex / triangle / group arrays as the reference's loader:
//   - every line starting with 'o' opens a group, "v " a vertex, "vn" a
//     normal, "f" a face of three or four "v/vt/vn" corners;
//   - a quad (a b c d) becomes triangles (a b c) and (d a c) with that
//     vertex order (rt_meshes.c:264-273);
//   - a face corner assigns its normal to the vertex it names (last wins);
//   - files without normals get them from rt_restore_normals;
//   - MTL: Ns -> specular.a, Ka/Kd/Ks -> ambient/diffuse/specular rgb,
//     Ni -> optDens; after the loader every material has color rgb = 1 and
//     reflect = 0.  As in the reference (rt_meshes.c:66-67) the opacity
//     color.a is taken from the number that follows the first two characters
//     of EVERY line that has one there, so the last such line wins (normally
//     the "d" line).
// Unlike the reference, arrays are zero-initialised and sized by their own
// element types, face corners may also be written "v", "v/vt" or "v//vn"
// (the reference's sscanf stops at the first corner of those and leaves the
// other indices uninitialised), negative indices count back from the last
// vertex / normal read so far, and a face that names a vertex outside the
// file's vertex list is dropped instead of being stored with a wild index.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "rt_meshes.h"

namespace {

bool starts( const char *s, const char *p ) { return strncmp( s, p, strlen( p ) ) == 0; }

// drops the line end and blanks before it (a name ends with its last visible character)
void chomp( char *s )
{
	size_t l = strlen( s );
	while ( l && ( s[l - 1] == '\n' || s[l - 1] == '\r' || s[l - 1] == ' ' || s[l - 1] == '\t' ) )
		s[--l] = '\0';
}

// number of whitespace-separated corners of a face line -> triangles
long face_triangles( const char *f )
{
	int words = 0;
	bool in = false;
	for ( ; *f; ++f )
	{
		bool sp = *f == ' ' || *f == '\t' || *f == '\n' || *f == '\r';
		if ( !sp && !in ) ++words;
		in = !sp;
	}
	return words == 3 ? 1 : words == 4 ? 2 : 0;
}

// one face corner: "v", "v/vt", "v//vn" or "v/vt/vn"; parts that are absent stay 0.
// Returns the position after the corner, NULL when no vertex index is there.
const char *parse_corner( const char *s, int &v, int &n )
{
	v = n = 0;
	while ( *s == ' ' || *s == '\t' ) ++s;
	char *e;
	long a = strtol( s, &e, 10 );
	if ( e == s )
		return NULL;
	v = (int) a;
	s = e;
	if ( *s == '/' )
	{
		++s;
		(void) strtol( s, &e, 10 );               // texture index, unused
		s = e;
		if ( *s == '/' )
		{
			++s;
			a = strtol( s, &e, 10 );
			if ( e != s )
				n = (int) a;
			s = e;
		}
	}
	return s;
}

char **load_mtl( rt_mesh *m, char *path )
{
	chomp( path );
	m->mc = 0;
	FILE *f = fopen( path, "r" );
	if ( !f )
		return NULL;
	char buf[255];
	while ( fgets( buf, sizeof( buf ), f ) )
		if ( starts( buf, "newmtl" ) )
			++m->mc;
	m->mat = (rt_material *) calloc( m->mc ? m->mc : 1, sizeof( rt_material ) );
	char **names = (char **) calloc( m->mc ? m->mc : 1, sizeof( char * ) );
	rewind( f );
	long cur = -1;
	while ( fgets( buf, sizeof( buf ), f ) )
	{
		if ( starts( buf, "newmtl" ) )
		{
			chomp( buf );
			++cur;
			names[cur] = strdup( strlen( buf ) > 7 ? buf + 7 : "" );
		}
		if ( cur < 0 )
			continue;
		rt_material &mt = m->mat[cur];
		if ( starts( buf, "newmtl" ) ) {}
		else if ( starts( buf, "Ns" ) ) sscanf( buf + 3, "%f", &mt.specular.a );
		else if ( starts( buf, "Ka" ) ) { sscanf( buf + 3, "%f %f %f", &mt.ambient.r, &mt.ambient.g, &mt.ambient.b ); mt.ambient.a = 0.0f; }
		else if ( starts( buf, "Kd" ) ) { sscanf( buf + 3, "%f %f %f", &mt.diffuse.r, &mt.diffuse.g, &mt.diffuse.b ); mt.diffuse.a = 0.0f; }
		else if ( starts( buf, "Ks" ) ) sscanf( buf + 3, "%f %f %f", &mt.specular.r, &mt.specular.g, &mt.specular.b );
		else if ( starts( buf, "Ni" ) ) sscanf( buf + 3, "%f", &mt.optDens );
		if ( strlen( buf ) > 2 )
			sscanf( buf + 2, "%f", &mt.color.a );
	}
	fclose( f );
	return names;
}

} // namespace

extern "C" {

void rt_restore_normals( rt_mesh *m )
{
	if ( !m )
		exit( -1 );
	for ( rt_ulong i = 0; i < m->tc; ++i )
	{
		if ( m->t[i].pV0 >= m->vc || m->t[i].pV1 >= m->vc || m->t[i].pV2 >= m->vc )
			continue;                                 // a wild index would be a wild write
		rt_vertex &a = m->v[m->t[i].pV0], &b = m->v[m->t[i].pV1], &c = m->v[m->t[i].pV2];
		rt_vector3 e1, e2, n;
		rt_vector3_sub( &a.pos, &b.pos, &e1 );        // b - a
		rt_vector3_sub( &a.pos, &c.pos, &e2 );        // c - a
		rt_vector3_cross( &e1, &e2, &n );
		rt_vector3_add( &a.norm, &n, &a.norm );
		rt_vector3_add( &b.norm, &n, &b.norm );
		rt_vector3_add( &c.norm, &n, &c.norm );
	}
	for ( rt_ulong i = 0; i < m->vc; ++i )
		rt_vector3_normalize( &m->v[i].norm, &m->v[i].norm );
}

int rt_mesh_load_from_obj( rt_mesh *m, char *path )
{
	if ( !m || !path )
		exit( -1 );
	FILE *f = fopen( path, "r" );
	if ( !f )
		return -1;
	char buf[255];
	rt_ulong nc = 0;
	char **names = NULL;
	m->gc = m->vc = m->tc = 0;
	m->mc = 0;
	m->mat = NULL;
	while ( fgets( buf, sizeof( buf ), f ) )
	{
		if ( starts( buf, "o" ) ) ++m->gc;
		else if ( starts( buf, "vn" ) ) ++nc;
		else if ( starts( buf, "vt" ) ) {}
		else if ( starts( buf, "v" ) ) ++m->vc;
		else if ( starts( buf, "f" ) ) m->tc += (rt_ulong) face_triangles( buf + 1 );
		else if ( starts( buf, "mtllib" ) && strlen( buf ) > 7 ) names = load_mtl( m, buf + 7 );
	}
	const bool implicitGroup = m->gc == 0;      // faces before any 'o' line still need a group
	if ( implicitGroup )
		m->gc = 1;
	m->v = (rt_vertex *) calloc( m->vc ? m->vc : 1, sizeof( rt_vertex ) );
	m->t = (rt_triangle *) calloc( m->tc ? m->tc : 1, sizeof( rt_triangle ) );
	m->g = (rt_mesh_group *) calloc( m->gc, sizeof( rt_mesh_group ) );
	rt_vector3 *n = nc ? (rt_vector3 *) calloc( nc, sizeof( rt_vector3 ) ) : NULL;
	rt_ulong curG = implicitGroup ? 1 : 0, curV = 0, curN = 0, curT = 0;

	auto corner = [&]( int vi, int ni ) {
		if ( n && ni >= 1 && (rt_ulong) ni <= nc && vi >= 1 && (rt_ulong) vi <= m->vc )
			m->v[vi - 1].norm = n[ni - 1];
	};
	auto valid = [&]( int vi ) { return vi >= 1 && (rt_ulong) vi <= m->vc; };
	auto emit = [&]( int a, int b, int c ) {
		if ( curT < m->tc && valid( a ) && valid( b ) && valid( c ) )
		{
			m->t[curT].pV0 = (rt_ulong) ( a - 1 );
			m->t[curT].pV1 = (rt_ulong) ( b - 1 );
			m->t[curT].pV2 = (rt_ulong) ( c - 1 );
			++curT;
		}
	};

	rewind( f );
	while ( fgets( buf, sizeof( buf ), f ) )
	{
		if ( starts( buf, "o" ) )
		{
			m->g[curG].voffset = curV;
			m->g[curG].toffset = curT;
			if ( curG > 0 )
			{
				m->g[curG - 1].vc = curV - m->g[curG - 1].voffset;
				m->g[curG - 1].tc = curT - m->g[curG - 1].toffset;
			}
			++curG;
		}
		else if ( starts( buf, "vn" ) )
		{
			if ( curN < nc )
			{
				sscanf( buf + 2, "%f %f %f", &n[curN].x, &n[curN].y, &n[curN].z );
				rt_vector3_normalize( n + curN, n + curN );
				++curN;
			}
		}
		else if ( starts( buf, "vt" ) ) {}
		else if ( starts( buf, "v" ) )
		{
			if ( curV < m->vc )
			{
				sscanf( buf + 2, "%f %f %f", &m->v[curV].pos.x, &m->v[curV].pos.y, &m->v[curV].pos.z );
				++curV;
			}
		}
		else if ( starts( buf, "f" ) )
		{
			chomp( buf );
			int vi[4] = { 0 }, ni[4] = { 0 };
			const long k = face_triangles( buf + 1 );
			const int corners = k == 1 ? 3 : ( k == 2 ? 4 : 0 );
			const char *p = buf + 1;
			bool ok = corners != 0;
			for ( int c = 0; c < corners && ok; ++c )
			{
				p = parse_corner( p, vi[c], ni[c] );
				ok = p != NULL;
				if ( ok && vi[c] < 0 ) vi[c] += (int) curV + 1;      // relative to the vertices read so far
				if ( ok && ni[c] < 0 ) ni[c] += (int) curN + 1;
			}
			if ( ok )
			{
				for ( int c = 0; c < corners; ++c ) corner( vi[c], ni[c] );
				emit( vi[0], vi[1], vi[2] );
				if ( k == 2 )
					emit( vi[3], vi[0], vi[2] );
			}
		}
		else if ( starts( buf, "usemtl" ) && curG > 0 && names && strlen( buf ) > 7 )
		{
			chomp( buf );
			for ( rt_ulong i = 0; i < m->mc; ++i )
				if ( names[i] && strcmp( buf + 7, names[i] ) == 0 )
					m->g[curG - 1].mat = m->mat + i;
		}
	}
	fclose( f );
	if ( curG > 0 )
	{
		m->g[curG - 1].vc = curV - m->g[curG - 1].voffset;
		m->g[curG - 1].tc = curT - m->g[curG - 1].toffset;
	}
	m->tc = curT;                                   // faces with wild indices were dropped
	for ( rt_ulong i = 0; i < m->mc; ++i )
	{
		m->mat[i].color.r = m->mat[i].color.g = m->mat[i].color.b = 1.0f;
		m->mat[i].reflect.r = m->mat[i].reflect.g = m->mat[i].reflect.b = 0.0f;
	}
	if ( n )
		free( n );
	else
		rt_restore_normals( m );
	if ( names )
	{
		for ( rt_ulong i = 0; i < m->mc; ++i ) free( names[i] );
		free( names );
	}
	return 0;
}

void rt_mesh_add( rt_render_pipe *pRp, rt_mesh *m, rt_ulong tg )
{
	if ( !pRp || !m )
		exit( -1 );
	rt_render_pipe_add_triangles( pRp, m->v, m->t, 0, 0, m->vc, m->tc, tg );
}

void rt_mesh_add_group( rt_render_pipe *pRp, rt_mesh *m, rt_ulong tg, rt_ulong g )
{
	if ( !pRp || !m )
		exit( -1 );
	rt_render_pipe_add_material( pRp, m->g[g].mat, tg );
	rt_render_pipe_add_triangles( pRp, m->v, m->t, m->g[g].voffset, m->g[g].toffset, m->g[g].vc, m->g[g].tc, tg );
}

void rt_mesh_reserve( rt_mesh *m, rt_ulong vc, rt_ulong tc, rt_ulong gc )
{
	if ( !m )
		exit( -1 );
	m->vc = vc;
	m->tc = tc;
	m->gc = gc;
	m->v = (rt_vertex *) calloc( vc ? vc : 1, sizeof( rt_vertex ) );
	m->t = (rt_triangle *) calloc( tc ? tc : 1, sizeof( rt_triangle ) );
	m->g = (rt_mesh_group *) calloc( gc ? gc : 1, sizeof( rt_mesh_group ) );
}

void rt_mesh_release( rt_mesh *m )
{
	if ( !m )
		exit( -1 );
	free( m->v ); m->v = NULL;
	free( m->t ); m->t = NULL;
	free( m->g ); m->g = NULL;
}

} // extern "C"
