/* oracle/lfold_cli.c -- TEST INFRASTRUCTURE. Reads FASTA on stdin, prints `RNALfold -L span` formatted text. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include "oracle.h"
int main(int argc, char **argv) {
    int span = 300, v185 = 0;
    for (int a = 1; a < argc; a++) {
        if (!strcmp(argv[a], "-L") && a + 1 < argc) span = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--vienna-1.8.5")) v185 = 1;
    }
    static char line[1 << 16];
    OracleFoldResult *R = malloc(sizeof(*R));
    while (fgets(line, sizeof line, stdin)) {
        size_t l = strlen(line);
        while (l && (line[l - 1] == '\n' || line[l - 1] == '\r')) line[--l] = 0;
        if (line[0] == '>') { printf("%s\n", line); continue; }
        if (!l) continue;
        int rc = v185 ? oracle_lfold185(line, (int)l, span, R) : oracle_lfold(line, (int)l, span, R);
        if (rc) { fprintf(stderr, "oracle_lfold failed rc=%d\n", rc); return 1; }
        for (int k = 0; k < R->n_lines; k++)
            printf("%s (%6.2f) %4d\n", R->lines[k].ss, R->lines[k].energy / 100., R->lines[k].start);
        for (size_t x = 0; x < l; x++) { char ch = toupper((unsigned char)line[x]); putchar(ch == 'T' ? 'U' : ch); }
        printf("\n (%6.2f)\n", R->mfe / 100.);
    }
    return 0;
}
