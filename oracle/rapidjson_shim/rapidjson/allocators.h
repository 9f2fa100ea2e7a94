// TEST INFRASTRUCTURE ONLY — see document.h in this directory.
#ifndef ORACLE_RAPIDJSON_SHIM_ALLOCATORS_H
#define ORACLE_RAPIDJSON_SHIM_ALLOCATORS_H
#include "rapidjson/document.h"
#endif
