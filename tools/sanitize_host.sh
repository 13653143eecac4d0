#!/bin/bash
# ThreadSanitizer and AddressSanitizer + UBSan over the host side of libacx (trie, failure links, flatten, the
# position-parallel section): builds a small harness outside the tree (/tmp) from acx_trie.cpp, acx_ppm.cpp,
# acx_items.cpp and runs make_automaton + flatten on 120,000 random signatures (bytes, then ACGT) with 8 host threads.
# No GPU, no HIP.  usage: tools/sanitize_host.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd); S=$R/pyahocorasick_amd/csrc; W=$(mktemp -d /tmp/acx_san.XXXX)
cat > $W/stub.cpp <<'EOC'
#include "acx_internal.h"
#include <cstdarg>
#include <cstdio>
int acx_fail(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); return code; }
EOC
gen() { cat > $W/main_$1.cpp <<EOC
#include "acx.h"
#include <cstdio>
#include <random>
int main() {
    acx_trie_t* t = nullptr; acx_trie_new(&t);
    std::mt19937_64 rng(5);
    for (int i = 0; i < 120000; i++) {
        uint8_t key[128]; int len = $2;
        for (int k = 0; k < len; k++) key[k] = (uint8_t)($3);
        int is_new = 0; acx_trie_add_word(t, key, (size_t)len, i, &is_new);
    }
    int changed = 0; int rc = acx_trie_make_automaton(t, &changed);
    void* blob = nullptr; size_t nbytes = 0; int rc2 = acx_flatten(t, &blob, &nbytes);
    printf("$1: rc %d %d, image %zu bytes\\n", rc, rc2, nbytes);
    acx_blob_free(blob); acx_trie_free(t);
    return rc | rc2;
}
EOC
}
gen bytes '4 + (int)(rng() % 60)' 'rng() & 0xFF'
gen dna '8 + (int)(rng() % 25)' '"ACGT"[rng() & 3]'
for san in thread address,undefined; do
  for w in bytes dna; do
    g++ -std=c++17 -O1 -g -fsanitize=$san -fno-sanitize-recover=undefined -I$R/include -I$S $W/main_$w.cpp $W/stub.cpp $S/acx_trie.cpp $S/acx_ppm.cpp $S/acx_items.cpp -o $W/t_$w -lpthread
    echo "== -fsanitize=$san, $w"; ACX_HOST_THREADS=8 $W/t_$w
  done
done
rm -rf $W
echo "sanitizers: clean"
