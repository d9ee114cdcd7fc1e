/*
 * ref_bench.c -- TEST INFRASTRUCTURE: a batch driver around the REAL reference (oracle/_ref/libmbe_ref.so), so that
 * bench.py's cpu_baseline leg can time the reference itself on the host cores of the GPU box without a Python call
 * per frame.  Built by `make -C oracle ref` into oracle/_ref/libref_bench.so (git-ignored, travels with the snapshot).
 * Streams run one after the other on the calling thread; the reference keeps its RNG per thread, so every stream is
 * re-seeded with mbe_setThreadRngSeed(seed0 + s) exactly as the fixtures do -- state parity is not the point here,
 * only the time per frame of the reference's own code path.
 */
#include <stddef.h>
#include <stdint.h>

#include "mbx_types.h" /* ABI-identical structs; reference prototypes restated below */

/* include/mbelib-neo/mbelib.h:433, 509, 568, 359, 596, 615 */
extern int mbe_processImbe7200x4400Frame(short*, mbe_process_result*, const char[8][23], char[88], mbe_parms*, mbe_parms*, mbe_parms*);
extern int mbe_processAmbe3600x2450Frame(short*, mbe_process_result*, const char[4][24], char[49], mbe_parms*, mbe_parms*, mbe_parms*);
extern int mbe_processImbe7100x4400Frame(short*, mbe_process_result*, const char[7][24], char[88], mbe_parms*, mbe_parms*, mbe_parms*);
extern int mbe_processAmbe3600x2400Frame(short*, mbe_process_result*, const char[4][24], char[49], mbe_parms*, mbe_parms*, mbe_parms*);
extern void mbe_setThreadRngSeed(uint32_t);

/* cells: S*T frames of `ncell` chars, stream-major; state: S*3 mbe_parms (in/out); pcm16: S*T*160 */
int
ref_process_batch(int codec, int S, int T, const char* cells, int ncell, mbe_parms* state, uint32_t seed0, int16_t* pcm16) {
    char bits[88];
    for (int s = 0; s < S; ++s) {
        mbe_parms* cur = &state[3 * (size_t)s];
        mbe_setThreadRngSeed(seed0 + (uint32_t)s);
        for (int t = 0; t < T; ++t) {
            const size_t f = (size_t)s * (size_t)T + (size_t)t;
            const char* fr = cells + f * (size_t)ncell;
            short* out = (short*)(pcm16 + f * 160);
            int rc;
            switch (codec) {
                case MBX_CODEC_IMBE7200X4400: rc = mbe_processImbe7200x4400Frame(out, NULL, (const char(*)[23])fr, bits, cur, cur + 1, cur + 2); break;
                case MBX_CODEC_AMBE3600X2450: rc = mbe_processAmbe3600x2450Frame(out, NULL, (const char(*)[24])fr, bits, cur, cur + 1, cur + 2); break;
                case MBX_CODEC_IMBE7100X4400: rc = mbe_processImbe7100x4400Frame(out, NULL, (const char(*)[24])fr, bits, cur, cur + 1, cur + 2); break;
                default: rc = mbe_processAmbe3600x2400Frame(out, NULL, (const char(*)[24])fr, bits, cur, cur + 1, cur + 2); break;
            }
            if (rc < 0) {
                return rc;
            }
        }
    }
    return 0;
}
