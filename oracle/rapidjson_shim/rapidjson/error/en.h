// TEST INFRASTRUCTURE ONLY — see ../document.h.
#ifndef ORACLE_RAPIDJSON_SHIM_ERROR_EN_H
#define ORACLE_RAPIDJSON_SHIM_ERROR_EN_H
#include "rapidjson/document.h"
namespace rapidjson {
inline const char *GetParseError_En(ParseErrorCode code) {
    return code == kParseErrorNone ? "No error." : "Invalid JSON syntax.";
}
}  // namespace rapidjson
#endif
