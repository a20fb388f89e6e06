// amx_plan.cpp - the planner alone (g++, no HIP): libamx_plan.so exports amx_model_plan_describe for tools/gen_specs.py.
#include "amx_plan.h"
extern "C" const char *amx_plan_last_error(void) { return g_err; }
