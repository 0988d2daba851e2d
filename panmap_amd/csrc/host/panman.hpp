// PanMAN (`.panman`) reader and node-genome materialiser.
//
// The reference loads a PanMAN through the external panman library (src/main.cpp:313-325) and
// materialises node genomes with panmapUtils::getStringFromReference (src/panmap_utils.cpp:7-190).
// This is an independent reader of the on-disk format (xz stream -> Cap'n Proto `TreeGroup`),
// laid out for the index builder: every block is flattened into a run of *columns*
// (for each consensus position: its gap columns, then the main base), all blocks concatenated in
// block-id order, so a node's genome is "the non-gap columns of the existing blocks".
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace pmx {

struct NucMut {
    int32_t block;
    int32_t pos;       // nucPosition
    int32_t gap;       // nucGapPosition or -1
    uint8_t len;       // mutInfo >> 4
    uint8_t type;      // mutInfo & 0xf
    uint32_t nucs;     // 24 bits, 4-bit codes left-aligned: i-th = (nucs >> 4*(5-i)) & 0xf
};

struct BlockMut {
    int32_t block;
    bool insertion;    // blockMutInfo
    bool inversion;
};

struct PanmanNode {
    std::string id;
    int32_t parent = -1;
    std::vector<int32_t> children;
    std::vector<BlockMut> block_muts;
    std::vector<NucMut> nuc_muts;
};

struct Panman {
    std::vector<PanmanNode> nodes;            // DFS pre-order == Newick pre-order; nodes[0] is the root
    // Column layout -------------------------------------------------------------------------
    int32_t n_blocks = 0;
    std::vector<int32_t> block_len;           // consensus length + 1 (trailing 'x' sentinel position)
    std::vector<uint32_t> block_col0;         // first column of block b; block_col0[n_blocks] == n_cols
    std::vector<std::vector<uint32_t>> pos_col;  // pos_col[b][p] = column of the main base of position p
    std::vector<std::vector<uint32_t>> gap_len;  // gap columns in front of position p
    std::vector<uint32_t> col_block;          // column -> block id
    uint32_t n_cols = 0;
    std::string consensus_cols;               // per column: base char, '-' for gap columns, 'x' sentinel

    // column of (block, nucPosition, nucGapPosition)
    inline int64_t column(int32_t b, int32_t pos, int32_t gap) const {
        if (b < 0 || b >= n_blocks || pos < 0 || pos >= block_len[b]) return -1;
        if (gap < 0) return pos_col[b][pos];
        if ((uint32_t)gap >= gap_len[b][pos]) return -1;
        return (int64_t)pos_col[b][pos] - gap_len[b][pos] + gap;
    }
    int32_t find_node(const std::string& id) const;
};

// Load `<path>.panman`. Throws std::runtime_error on malformed input.
void load_panman(const std::string& path, Panman& out);
// Parse an already-decompressed Cap'n Proto stream (for tests).
void parse_panman(const uint8_t* buf, size_t len, Panman& out);

// xz decompression through liblzma.so.5 (dlopen; the image ships no lzma headers)
std::vector<uint8_t> xz_decompress(const uint8_t* in, size_t n);

// 4-bit code -> nucleotide char (panmanUtils::getNucleotideFromCode)
char nuc_from_code(int code);
char complement_iupac(char c);

// Mutable per-DFS state: the column characters and block on/off/strand flags of the current node.
struct PanmanState {
    std::string cols;                 // current char per column ('-' gap, 'x' sentinel)
    std::vector<uint8_t> block_exists;
    std::vector<uint8_t> block_fwd;   // strand: 1 forward, 0 inverted
    void init(const Panman& pm);
};

struct UndoLog {
    std::vector<std::pair<uint32_t, char>> col_changes;                 // (column, old char)
    std::vector<std::pair<int32_t, std::pair<uint8_t, uint8_t>>> blocks;  // (block, old exists, old fwd)
};

// One recorded mutated column range [a,b] (src/panmap_utils.hpp:783-842: recorded even when the new
// base equals the old one; a block mutation records the whole block).
struct ColRange { uint32_t a, b; };

// Apply node `ni`'s block then nuc mutations (src/panmap_utils.hpp:726-842, src/panmap_utils.cpp:92-131).
void apply_node(const Panman& pm, int32_t ni, PanmanState& st, UndoLog* undo, std::vector<ColRange>* ranges);
void undo_node(PanmanState& st, const UndoLog& undo);

// Ungapped genome of the current state (src/panmap_utils.cpp:134-180, aligned=false).
std::string genome_of_state(const Panman& pm, const PanmanState& st);
// Genome of node `ni` (walks root -> node).  root_state: NULL, or the state after the root's own mutations (the same for
// every node of a tree, and most of the work: the root turns the consensus into a genome) -- the walk then starts there.
std::string node_genome(const Panman& pm, int32_t ni, const PanmanState* root_state = nullptr);
PanmanState root_state_of(const Panman& pm);

}  // namespace pmx
