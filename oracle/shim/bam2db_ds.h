/* bam2db_ds.h — the two-line shim INTEGRATION.md section 2 prescribes: the reference's main.c includes
 * "bam2db_ds.h" for `int bam2db(...)` and `extern int _umi_copies_flag` (bam2db_ds.h:23,62-70); with the MI355X
 * engine linked in, those come from fastf_amd.h.  Used by `make -C oracle refcli` (link test of the drop-in). */
#include "fastf_amd.h"
