// Part 1 of attention.hip: attn_dkv_kernel<96, false, false> (the FAST dK/dV kernel) and its launcher, built with its own flags
// (Makefile: FLAGS_attention_dkv).  See the note at the top of attention.hip.
#define CSTS_ATTN_PART 1
#include "attention.hip"
