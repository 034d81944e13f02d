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
            fastf_process_is_exiting_ = getenv("FASTF_FULL_TEARDOWN") == NULL;   /* this process ends with the command */
            int rc = commands[i].fn(argc - 1, argv + 1);
            /* every output is closed by now; skip the ~0.2 s the HIP runtime spends unloading at exit */
            fflush(NULL);
            _exit(rc);
        }
    fprintf(stderr, "\x1b[31mError:\x1b[0m unknown command `%s`\n", argv[1]);
    return 1;
}
