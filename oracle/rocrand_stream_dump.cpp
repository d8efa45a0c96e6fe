// TEST INFRASTRUCTURE.  Prints rocRAND's Philox4x32-10 host-side stream so that
// tests/test_philox.py can check oracle_draw_block() against it word for word.
// usage: rocrand_stream_dump <seed> <subsequence> <first_block> <nblocks>
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_philox4x32_10.h>
#include <cstdio>
#include <cstdlib>

int main(int argc, char** argv)
{
    if (argc != 5) return 2;
    unsigned long long seed = strtoull(argv[1], nullptr, 0);
    unsigned long long sub  = strtoull(argv[2], nullptr, 0);
    unsigned long long blk  = strtoull(argv[3], nullptr, 0);
    int n = atoi(argv[4]);
    rocrand_state_philox4x32_10 st;
    rocrand_init(seed, sub, 4ull * blk, &st);
    for (int i = 0; i < n; ++i) {
        uint4 w = rocrand4(&st);
        printf("%u %u %u %u\n", w.x, w.y, w.z, w.w);
    }
    return 0;
}
