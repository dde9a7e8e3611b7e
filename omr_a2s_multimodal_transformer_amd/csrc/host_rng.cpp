// Token noise of Transformer.apply_teacher_forcing (reference: src/transformer/model.py:152-160) as ONE host call: the
// reference walks the B x T token matrix in Python and, per token, draws random.random() and -- for a replaced token only --
// random.randint(0, V - 1) from Python's global Mersenne Twister.  The draw sequence is part of the behaviour (golden F14), so
// this function continues the interpreter's OWN generator: the caller passes random.getstate()'s 624 words + index, the loop
// below makes exactly the calls CPython's `random` module would (Modules/_randommodule.c: random() = (a * 2^26 + b) / 2^53
// with a = genrand >> 5, b = genrand >> 6; randint -> randrange -> _randbelow_with_getrandbits: k = n.bit_length(),
// r = genrand >> (32 - k) until r < n), and hands the advanced state back for random.setstate().
#include <cstdint>

#include "omr_hip.h"

namespace {
constexpr int N = 624, M = 397;
inline uint32_t genrand(uint32_t* mt, int& mti) {
    if (mti >= N) {                                     // MT19937 regeneration (Matsumoto & Nishimura, 1998)
        static const uint32_t mag01[2] = {0x0u, 0x9908b0dfu};
        int kk = 0;
        for (; kk < N - M; ++kk) { const uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu); mt[kk] = mt[kk + M] ^ (y >> 1) ^ mag01[y & 1u]; }
        for (; kk < N - 1; ++kk) { const uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu); mt[kk] = mt[kk + (M - N)] ^ (y >> 1) ^ mag01[y & 1u]; }
        const uint32_t y = (mt[N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ mag01[y & 1u];
        mti = 0;
    }
    uint32_t y = mt[mti++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}
}  // namespace

extern "C" int omr_teacher_forcing_noise(long* tokens, long count, double prob, long pad_idx, long vocab, unsigned int* mt_state, int* mt_index) {
    if (!tokens || !mt_state || !mt_index || count < 0 || vocab < 1 || vocab > (1L << 31) || *mt_index < 0 || *mt_index > N) return -1;
    int k = 0;
    while ((vocab >> k) != 0) ++k;                      // vocab.bit_length()
    int mti = *mt_index;
    for (long i = 0; i < count; ++i) {
        const uint32_t a = genrand(mt_state, mti) >> 5, b = genrand(mt_state, mti) >> 6;
        const double r = (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
        if (r < prob && tokens[i] != pad_idx) {         // the reference's short-circuit: randint is drawn for replaced tokens only
            uint32_t v = genrand(mt_state, mti) >> (32 - k);
            while ((long)v >= vocab) v = genrand(mt_state, mti) >> (32 - k);
            tokens[i] = (long)v;
        }
    }
    *mt_index = mti;
    return 0;
}
