/*
 * oracle_cli.c -- TEST INFRASTRUCTURE. Small driver around llama2_oracle.c:
 *   oracle_cli synth <dim> <hidden> <layers> <heads> <kv_heads> <vocab(+/-)> <seq_len> <seed> <out.bin>
 *   oracle_cli run   <ckpt.bin> <steps> <logits_out.f32|-> <tokens_out.i32|->   (greedy from BOS, llama2.ts:463-478)
 *   oracle_cli sample <ckpt.bin> <steps> <temperature> <topp> <seed>            (sampled from BOS, llama2.ts:476-493; prints ids)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "llama2_oracle.h"

int main(int argc, char** argv) {
  if (argc >= 11 && !strcmp(argv[1], "synth")) {
    int32_t hdr[7];
    for (int i = 0; i < 7; ++i) hdr[i] = atoi(argv[2 + i]);
    return orc_synth_write(hdr, (uint32_t)strtoul(argv[9], NULL, 10), argv[10]) ? 1 : 0;
  }
  if (argc >= 6 && !strcmp(argv[1], "run")) {
    orc_model* m = orc_open(argv[2]);
    if (!m) { fprintf(stderr, "cannot open %s\n", argv[2]); return 1; }
    const int steps = atoi(argv[3]);
    const int V = orc_get_config(m)->vocab_size;
    FILE* fl = strcmp(argv[4], "-") ? fopen(argv[4], "wb") : NULL;
    FILE* ft = strcmp(argv[5], "-") ? fopen(argv[5], "wb") : NULL;
    float* logits = (float*)malloc((size_t)V * 4);
    int32_t token = 1;
    for (int pos = 0; pos < steps; ++pos) {
      orc_forward(m, token, pos, logits);
      if (fl) fwrite(logits, 4, (size_t)V, fl);
      if (ft) fwrite(&token, 4, 1, ft);
      token = orc_argmax(logits, V);
    }
    if (fl) fclose(fl);
    if (ft) fclose(ft);
    free(logits);
    orc_destroy(m);
    return 0;
  }
  if (argc >= 7 && !strcmp(argv[1], "sample")) {
    orc_model* m = orc_open(argv[2]);
    if (!m) { fprintf(stderr, "cannot open %s\n", argv[2]); return 1; }
    const int steps = atoi(argv[3]);
    const double temperature = atof(argv[4]), topp = atof(argv[5]);
    uint64_t rng = strtoull(argv[6], NULL, 10);
    const int V = orc_get_config(m)->vocab_size;
    float* logits = (float*)malloc((size_t)V * 4);
    int32_t token = 1;
    for (int pos = 0; pos < steps; ++pos) {
      orc_forward(m, token, pos, logits);
      token = orc_next_token(logits, V, temperature, topp, &rng);
      printf("%d%c", token, pos + 1 == steps ? '\n' : ' ');
    }
    free(logits);
    orc_destroy(m);
    return 0;
  }
  fprintf(stderr, "usage: see oracle_cli.c\n");
  return 2;
}
