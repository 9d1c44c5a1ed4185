// Independent pin of the k-NN row ORDER (test infrastructure; built with g++ by tests/test_heap_pin.py).
//
// The reference's selectKNearestRoadEntities (reference src/knn.hpp:103-158) drives a vendored copy of the SGI STL
// heap (src/binary_heap.hpp).  libstdc++'s std::make_heap / std::pop_heap / std::push_heap are a third-party
// implementation of the same SGI algorithm, so running knn.hpp's loop on them gives an order that depends on
// neither the oracle's restatement of binary_heap.hpp (oracle/gd_oracle.c) nor the HIP kernel.
//
// stdin (binary): int32 K, int32 R, float radius, then R float32 keys (position.length2() of every road's
// observation, in road order).  stdout (binary): K int32 = road index per output slot, -1 for a zero-filled slot.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

struct Obs {
    float key;  // position.length2()
    int32_t road;
};
static bool cmp(const Obs &l, const Obs &r) { return l.key < r.key; }  // knn.hpp:15-17

// knn.hpp:83-97 (length() = sqrtf(length2()))
static long radius_filter(Obs *heap, long K, float radius) {
    long new_beyond = K, idx = 0;
    while (idx < new_beyond) {
        if (std::sqrt(heap[idx].key) <= radius) { ++idx; continue; }
        heap[idx] = heap[--new_beyond];
    }
    return new_beyond;
}

int main() {
    int32_t K = 0, R = 0;
    float radius = 0;
    if (std::fread(&K, 4, 1, stdin) != 1 || std::fread(&R, 4, 1, stdin) != 1 || std::fread(&radius, 4, 1, stdin) != 1) return 2;
    std::vector<float> keys(R);
    if (R && std::fread(keys.data(), 4, R, stdin) != (size_t)R) return 2;
    std::vector<Obs> heap(K);
    const long first = std::min<long>(R, K);
    for (long i = 0; i < first; i++) heap[i] = Obs{keys[i], (int32_t)i};  // :112-120
    long beyond;
    if (R < K) {
        beyond = radius_filter(heap.data(), R, radius);  // :122-126
    } else {
        std::make_heap(heap.begin(), heap.end(), cmp);  // :128
        for (long r = K; r < R; r++) {                  // :130-151
            const Obs cur{keys[r], (int32_t)r};
            if (!cmp(cur, heap[0])) continue;
            std::pop_heap(heap.begin(), heap.end(), cmp);
            heap[K - 1] = cur;
            std::push_heap(heap.begin(), heap.end(), cmp);
        }
        beyond = radius_filter(heap.data(), K, radius);  // :156
    }
    std::vector<int32_t> out(K, -1);
    for (long i = 0; i < beyond; i++) out[i] = heap[i].road;
    std::fwrite(out.data(), 4, K, stdout);
    return 0;
}
