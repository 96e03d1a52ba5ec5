#include "lizard_oracle.h"
size_t lzo_huf_compress(void* dst, size_t dstCapacity, const void* src, size_t srcSize)
{ (void)dst; (void)dstCapacity; (void)src; (void)srcSize; return dst ? (size_t)-1 : (size_t)-2; }
