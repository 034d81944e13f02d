/* extract.h — shim of INTEGRATION.md section 2: main.c and count.h include "extract.h" for the crb / extract
 * prototypes (extract.h:15-26); libfastf_amd.so exports them, fastf_amd.h declares them (the tree types are the
 * reference's own, taken from its filter.h when that was included first). */
#ifndef FASTF_EXTRACT_SHIM_H
#define FASTF_EXTRACT_SHIM_H
#include "filter.h"
#include "fastf_amd.h"
void print_CB_node(CB_node *root, gzFile fp);
#endif
