/* oracle_cli.c -- command-line front end of the CPU oracle (test infrastructure only).
 * usage: oracle_cli <bedGraph> <penalty> <db>   (exit status = solver status code) */
#include <stdio.h>
#include "peakseg_oracle.h"
int main(int argc, char **argv) {
  if (argc != 4) {
    fprintf(stderr, "usage: %s bedGraph penalty db   [math=%s]\n", argv[0], oracle_math_kind());
    return 64;
  }
  return oracle_PeakSegFPOP_disk(argv[1], argv[2], argv[3]);
}
