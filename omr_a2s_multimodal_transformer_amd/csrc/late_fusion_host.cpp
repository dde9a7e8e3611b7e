// Host side of the late (prediction-level) fusion: Smith-Waterman local alignment of two token-id sequences.
// Reference: src/multimodal/smith_waterman/test.py:136-150 calls swalign.LocalAlignment(NucleotideScoringMatrix(match,
// mismatch), gap_penalty).align(ref, query).  swalign (0.3.x) is a third-party package that is absent from the reference
// tree and from this image: this file restates its published dynamic programme -- cell = max(diag + score, up + gap,
// left + gap, 0) with the extension penalty (default -1) for a cell that continues a gap run, "prefer gap runs" tie
// order (continue deletion, continue insertion, match, new deletion, new insertion), the LAST maximal cell as the end
// point, traceback until a non-positive cell -- PARITY UNPINNED (no fixture of the package's output exists to pin it).
// O(nq * nr) integer work on the host, tokens compared as integers (the reference maps tokens to characters first).
#include <cstddef>
#include <vector>

#include "omr_hip.h"

#define OMR_ERR_ARG (-1)

extern "C" int omr_sw_align(const int* ref, int nr, const int* query, int nq, int match, int mismatch, int gap_penalty,
                            int gap_extension_penalty, char* ops, int* r_pos, int* q_pos, int* score) {
    if (!ref || !query || nr <= 0 || nq <= 0 || !ops || !r_pos || !q_pos) return OMR_ERR_ARG;
    const int W = nr + 1;
    std::vector<int> val((size_t)(nq + 1) * W, 0), run((size_t)(nq + 1) * W, 0);
    std::vector<char> op((size_t)(nq + 1) * W, ' ');
    for (int row = 1; row <= nq; ++row) op[(size_t)row * W] = 'i';
    for (int col = 1; col <= nr; ++col) op[col] = 'd';
    int max_val = 0, max_row = 0, max_col = 0;
    for (int row = 1; row <= nq; ++row)
        for (int col = 1; col <= nr; ++col) {
            const size_t c = (size_t)row * W + col, up = c - W, left = c - 1, diag = c - W - 1;
            const int mm_val = val[diag] + (query[row - 1] == ref[col - 1] ? match : mismatch);
            int ins_run = 0, del_run = 0, ins_val, del_val;
            if (op[up] == 'i') { ins_run = run[up]; ins_val = val[up] == 0 ? 0 : val[up] + gap_extension_penalty; }
            else ins_val = val[up] + gap_penalty;
            if (op[left] == 'd') { del_run = run[left]; del_val = val[left] == 0 ? 0 : val[left] + gap_extension_penalty; }
            else del_val = val[left] + gap_penalty;
            int cell = mm_val;
            if (del_val > cell) cell = del_val;
            if (ins_val > cell) cell = ins_val;
            if (cell < 0) cell = 0;
            char o; int rl;
            if (del_run && cell == del_val) { o = 'd'; rl = del_run + 1; }
            else if (ins_run && cell == ins_val) { o = 'i'; rl = ins_run + 1; }
            else if (cell == mm_val) { o = 'm'; rl = 0; }
            else if (cell == del_val) { o = 'd'; rl = 1; }
            else if (cell == ins_val) { o = 'i'; rl = 1; }
            else { o = 'x'; rl = 0; cell = 0; }
            val[c] = cell; op[c] = o; run[c] = rl;
            if (cell >= max_val) { max_val = cell; max_row = row; max_col = col; }
        }
    int row = max_row, col = max_col, n = 0;
    std::vector<char> rev;
    while (true) {
        const size_t c = (size_t)row * W + col;
        if (val[c] <= 0) break;
        const char o = op[c];
        rev.push_back(o);
        if (o == 'm') { --row; --col; }
        else if (o == 'i') --row;
        else if (o == 'd') --col;
        else break;
    }
    n = (int)rev.size();
    for (int i = 0; i < n; ++i) ops[i] = rev[n - 1 - i];
    *r_pos = col; *q_pos = row;
    if (score) *score = max_val;
    return n;                      // number of alignment columns (<= nr + nq); ops[i] in {'m', 'i', 'd'}
}
