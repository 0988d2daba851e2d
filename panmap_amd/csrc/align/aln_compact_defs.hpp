// Capacities of the compact align tier (aln_compact.hpp), shared with the host launcher.
#pragma once
#define PMX_C_CAP 48            // seeds / anchors per pair
#define PMX_C_MCAP 40           // minimizers of one read waiting for their probes
#define PMX_C_NW 5              // 32-base words per read: reads up to 160 bases
#define PMX_C_MAXLEN (32 * PMX_C_NW)
#define PMX_C_LANE_WORDS 156    // 624 bytes per pair
