// bsvi_internal.h — shared between the translation units of libbsvi.so (not part of the C ABI).
#pragma once
#include <string>

// records the thread-local message returned by bsvi_last_error() and returns `code`
int bsvi_fail(int code, const std::string& msg);
