#include <cstdio>
#include <cstdlib>
#include "ahsoka_phase.h"
int main(int argc, char** argv) {
  const char* stems[] = {"toy", "c1", "multi", "sbub"};
  for (const char* s : stems) {
    char g[256], a[256];
    snprintf(g, sizeof g, "tests/golden/%s.gfa", s);
    snprintf(a, sizeof a, "tests/golden/%s.gaf", s);
    ahsoka_chain_batch* b = nullptr;
    int rc = ahsoka_ingest(g, a, &b);
    printf("%s ingest rc=%d chains=%d\n", s, rc, b ? b->n_chains : -1);
    char out[256];
    snprintf(out, sizeof out, "/tmp/san_%s.dump", s);
    rc = ahsoka_ingest_dump(g, a, out);
    printf("  dump rc=%d\n", rc);
    if (b) ahsoka_batch_host_free(b);
  }
  ahsoka_synth_spec sp{200, 12, 120, 2, 1, 10.0, 0.02, 42ull, 0};
  ahsoka_chain_batch* b = nullptr;
  printf("synth rc=%d\n", ahsoka_synth_batch(&sp, &b));
  if (b) ahsoka_batch_host_free(b);
  printf("write rc=%d\n", ahsoka_synth_write_text(&sp, "/tmp/san_x.gfa", "/tmp/san_x.gaf"));
  b = nullptr;
  printf("ingest synth rc=%d\n", ahsoka_ingest("/tmp/san_x.gfa", "/tmp/san_x.gaf", &b));
  if (b) { printf("chains=%d\n", b->n_chains); ahsoka_batch_host_free(b); }
  // malformed inputs
  FILE* f = fopen("/tmp/san_bad.gfa", "w"); fputs("S\tutg1l\tACGT\nL\tutg1l\t+\tutgXl\t-\t0M\nS\nL\t\n", f); fclose(f);
  f = fopen("/tmp/san_bad.gaf", "w"); fputs("r1\t10\t0\t10\t+\t>utg1l<utg9l\n\n\t\t\n", f); fclose(f);
  b = nullptr;
  printf("bad ingest rc=%d\n", ahsoka_ingest("/tmp/san_bad.gfa", "/tmp/san_bad.gaf", &b));
  if (b) ahsoka_batch_host_free(b);
  return 0;
}
