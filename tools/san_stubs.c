/* the sanitizer harnesses' stubs (tools/san_stubs.h) as a translation unit of its own, for harnesses written in C++ */
#include "san_stubs.h"
