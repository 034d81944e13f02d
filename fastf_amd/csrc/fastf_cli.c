/* fastf_cli.c — `fastF` command line: dispatch table of the reference (main.c:404-443),
 * with the subcommands this engine implements (the BAM ones: bam2db, crb, extract). */
#include "fastf_amd.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

extern int fastf_process_is_exiting_;
struct cmd_struct { const char *cmd; int (*fn)(int, const char **); };
static const struct cmd_struct commands[] = { {"crb", cmd_crb}, {"bam2db", cmd_bam2db}, {"extract", cmd_extract} };

/* One command uses one GPU unless FASTF_DEVICES asks for more: hide the others from the HIP runtime before it loads,
 * so that its start-up (which opens every visible device) costs the same on an 8-GPU host as on a 1-GPU one.
 * Any visibility setting of the caller's wins; FASTF_ALL_DEVICES=1 switches this off. */
static void narrow_visible_devices(void)
{
    if (getenv("FASTF_DEVICES") || getenv("FASTF_ALL_DEVICES") || getenv("ROCR_VISIBLE_DEVICES") ||
        getenv("HIP_VISIBLE_DEVICES") || getenv("CUDA_VISIBLE_DEVICES") || getenv("GPU_DEVICE_ORDINAL")) return;
    const char *dev = getenv("FASTF_DEVICE");
    if (dev && (dev[0] < '0' || dev[0] > '9' || strlen(dev) > 3)) return;
    setenv("ROCR_VISIBLE_DEVICES", dev ? dev : "0", 1);
    setenv("FASTF_DEVICE", "0", 1);
}

int main(int argc, const char **argv)
{
    if (argc < 2 || !strcmp(argv[1], "-h") || !strcmp(argv[1], "--help")) {
        printf("Usage: fastF <command> [options]\n\nCommands:\n"
               "    crb       CB/CR tags of a BAM and their frequencies -> gzip'ed rows\n"
               "    bam2db    BAM -> down-sampled, UMI-deduplicated gene x cell matrix (MI355X engine)\n"
               "    extract   frequencies of one BAM tag -> tag_summary.csv\n\n"
               "(%s; the FASTQ commands freq/filter are not part of this engine)\n", fastf_version());
        return argc < 2 ? 1 : 0;
    }
    for (size_t i = 0; i < sizeof commands / sizeof commands[0]; i++)
        if (!strcmp(commands[i].cmd, argv[1])) {
            narrow_visible_devices();
            fastf_process_is_exiting_ = getenv("FASTF_FULL_TEARDOWN") == NULL;   /* this process ends with the command */
            int rc = commands[i].fn(argc - 1, argv + 1);
            /* every output is closed by now; skip the ~0.2 s the HIP runtime spends unloading at exit */
            fflush(NULL);
            if (!fastf_process_is_exiting_) return rc;       /* FASTF_FULL_TEARDOWN: an ordinary exit (a profiler writes its output from exit handlers) */
            _exit(rc);
        }
    fprintf(stderr, "\x1b[31mError:\x1b[0m unknown command `%s`\n", argv[1]);
    return 1;
}
