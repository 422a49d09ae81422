/* Experiment only: LD_PRELOAD bump allocator (free is a no-op) to see how much of a host leg is malloc/free.
 *   gcc -O2 -shared -fPIC -o bump_malloc.so bump_malloc.c */
#define _GNU_SOURCE
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <sys/mman.h>
static __thread char* cur; static __thread char* lim;
static void* grab(size_t n) {
    n = (n + 15) & ~(size_t)15;
    size_t need = n + 16;
    if (need > (size_t)(lim - cur)) {
        size_t chunk = need > (256u << 20) ? need : (256u << 20);
        char* p = mmap(0, chunk, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p == MAP_FAILED) return 0;
        cur = p; lim = p + chunk;
    }
    *(size_t*)cur = n;
    void* r = cur + 16;
    cur += need;
    return r;
}
void* malloc(size_t n) { return grab(n); }
void free(void* p) { (void)p; }
void* calloc(size_t a, size_t b) { return grab(a * b); /* fresh mmap pages are zero; never reused */ }
void* realloc(void* p, size_t n) {
    if (!p) return grab(n);
    size_t old = *(size_t*)((char*)p - 16);
    if (n <= old) return p;
    void* q = grab(n);
    if (q) memcpy(q, p, old);
    return q;
}
void* memalign(size_t al, size_t n) { char* p = grab(n + al); return (void*)(((uintptr_t)p + al - 1) & ~(uintptr_t)(al - 1)); }
int posix_memalign(void** out, size_t al, size_t n) { *out = memalign(al, n); return *out ? 0 : 12; }
void* aligned_alloc(size_t al, size_t n) { return memalign(al, n); }
