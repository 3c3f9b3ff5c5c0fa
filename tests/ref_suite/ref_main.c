/* ref_main.c — process wrapper for the reference's test programs (tests/ref_suite/Makefile compiles each of them
 * with -Dmain=ref_test_main).  Without arguments, or with the program's own arguments, it simply calls the test's
 * main.  With "--csv FILE [stride]" it calls the test's main once per row of the reference's parameter list (the
 * rows CMake turns into one ctest case each, tests/CMakeLists.txt:66-100), in ONE process: a HIP process start per
 * row would cost minutes of GPU-box time.  Exit code: number of failing rows (capped at 255). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

extern int ref_test_main(int argc, char** argv);

int main(int argc, char** argv) {
  if (argc >= 3 && strcmp(argv[1], "--csv") == 0) {
    FILE* f = fopen(argv[2], "r");
    const int stride = argc >= 4 ? atoi(argv[3]) : 1;
    char line[512];
    int row = 0, failed = 0, ran = 0;
    if (!f) { fprintf(stderr, "cannot open %s\n", argv[2]); return 255; }
    if (!fgets(line, sizeof line, f)) { fclose(f); return 255; }   /* header row */
    while (fgets(line, sizeof line, f)) {
      char* args[16];
      int n = 1;
      char* tok;
      if (row++ % (stride > 0 ? stride : 1)) continue;
      line[strcspn(line, "\r\n")] = 0;
      if (!line[0]) continue;
      args[0] = argv[0];
      for (tok = strtok(line, ","); tok && n < 15; tok = strtok(NULL, ",")) args[n++] = tok;
      args[n] = NULL;
      ran++;
      if (ref_test_main(n, args) != 0) {
        int k;
        failed++;
        fprintf(stderr, "\nFAILED row %d:", row);
        for (k = 1; k < n; k++) fprintf(stderr, " %s", args[k]);
        fprintf(stderr, "\n");
      }
    }
    fclose(f);
    printf("\n%s: %d rows run, %d failed\n", argv[2], ran, failed);
    return failed > 255 ? 255 : failed;
  }
  return ref_test_main(argc, argv);
}
