/* Umbrella header, same role as the reference's rt.h:1-2. */
#ifndef RT_H
#define RT_H
#include "rt_funcs_primitives.h"
#include "rt_funcs_render_pipe.h"
#endif
