// Shared by the translation units of the C-ABI: error reporting (message kept per thread).
#pragma once
int csmpn_fail(int code, const char* fmt, ...);
