// Capacities of the compact align tier (aln_compact.hpp), shared with the host launcher.
#pragma once
#define PMX_C_W 11              // minimizer window of the tier's sketch (the preset's, src/mm_align.c:140-166; k must be odd)
#define PMX_C_CAP 56            // seeds / anchors per pair: six bytes each: the most that leaves seven waves per CU (0.1 % of 150 bp pairs -- mean 40
                                // anchors -- run out; with 48, 0.9 % did and a tier of bails costs more than that; at most 63, see CMemT)
#define PMX_C_MCAP 40           // minimizers of one read waiting for their probes
#define PMX_C_SEEDQ 14           // k_compact_seeds: minimizers of one read waiting for their probes (LDS: 112 bytes per pair)
#define PMX_C_NW 5              // 32-base words per read: reads up to 160 bases
#define PMX_C_MAXLEN (32 * PMX_C_NW)
// LDS words per pair (CMemT<PT>::kWords): X (PT) + Y (u16) + G (u16), PMX_C_CAP entries each
#define PMX_C_LANE_WORDS16 84    // 336 bytes: reference position words of 16 bits
#define PMX_C_LANE_WORDS32 112   // 448 bytes
// The chain kernel's first form (k_align_compact16/32: seeds from the hand-over) keeps 48 anchors: 288 bytes per pair, 19.6 KB
// per wave with the penalty tables, EIGHT waves per CU instead of seven (nine with 32-bit positions... six instead of five).
// 0.9 % of 150 bp pairs have 49 .. 56 seeds (mean 39.9, sd 3.9): those go to the second form, whose memory holds all 56.
#define PMX_C_CAP1 48
#define PMX_C_LANE_WORDS16_1 72  // 288 bytes
#define PMX_C_LANE_WORDS32_1 96  // 384 bytes
// gap-penalty tables of the chain fill, per wave (aln_compact.hpp CPenTab)
#define PMX_C_PEN_SAME 128
#define PMX_C_PEN_DIFF 1024
#define PMX_C_PEN_BYTES (PMX_C_PEN_SAME + PMX_C_PEN_DIFF)
// second form of the tier (several regions per mate, aln_compact_multi.hpp): its workspace in HBM
#define PMX_CM_MAXC 4            // fragment chains of a pair (and therefore regions per mate) this form follows
#define PMX_CM_SREGS (4 * PMX_CM_MAXC)   // region records per pair: fragment chains, mate 0, mate 1, the sorter's copy (24 words each)
#define PMX_CM_INTS 64           // index words per pair
#define PMX_CM_WS_WORDS (PMX_CM_SREGS * 24 + PMX_CM_INTS)   // 4-byte words per pair (x 64 lanes per wave: 114,688 bytes)
