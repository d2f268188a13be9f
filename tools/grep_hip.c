/* grep_hip.c - the reference's examples/grep.rs:42-56 in plain C over the C ABI:
 *     ./grep_hip <needle> <file>
 * builds one searcher (position = the `new` default, last byte) and runs ONE search_in over the file.
 *   gcc -O2 -I include tools/grep_hip.c -o grep_hip -L sliceslice-rs_amd/csrc -lsliceslice_hip \
 *       -Wl,-rpath,$PWD/sliceslice-rs_amd/csrc
 */
#include <stdio.h>
#include <string.h>

#include "sliceslice_hip.h"

int main(int argc, char **argv)
{
    if (argc != 3) {
        fprintf(stderr, "./grep_hip <needle> <file>\n");
        return 2;
    }
    ss_searcher *s = NULL;
    int found = 0;
    if (ss_searcher_new((const uint8_t *)argv[1], strlen(argv[1]), &s) != SS_OK ||
        ss_search_file(s, argv[2], &found) != SS_OK) {
        fprintf(stderr, "error: %s\n", ss_last_error());
        ss_searcher_free(s);
        return 2;
    }
    printf("Searching for %s in \"%s\": %s\n", argv[1], argv[2], found ? "true" : "false");
    ss_searcher_free(s);
    return 0;
}
