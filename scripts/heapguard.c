/* Development aid: a malloc interposer (LD_PRELOAD) that finds WRITES INTO FREED MEMORY in a process whose libraries cannot be
 * rebuilt with a sanitizer (HIP's runtime, torch, libgomp beside this library and the oracle: profiles/r06_segv_hunt.txt).
 *
 * free() / operator delete do not free: blocks of 24 .. 4096 usable bytes are filled with 0xC5 and parked in a first-in first-out
 * quarantine (1 M blocks / 256 MB); when a block leaves it — and for all that remain at exit — every byte is compared with the fill.
 * A changed byte is a write after free: the report names the block's size, the offset and the bytes written, and WHO FREED IT (the
 * return address of the free / delete, as library + offset; operator delete is interposed as well so that the address lies in the
 * deleting destructor, which names the class).  Reports go to stderr and to $HEAPGUARD_LOG.<pid>.
 *
 * build: gcc -O2 -fPIC -shared -o heapguard.so heapguard.c -ldl     use: LD_PRELOAD=.../heapguard.so python ...   */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <fcntl.h>
#include <malloc.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define FILL 0xC5
#define MIN_SIZE 24
#define MAX_SIZE 4096
#define RING (1u << 20)
#define MAX_BYTES (256ll << 20)

static void* (*real_malloc)(size_t);
static void (*real_free)(void*);
static void* (*real_calloc)(size_t, size_t);
static void* (*real_realloc)(void*, size_t);

static char boot[1 << 16];
static size_t boot_used;
static int resolving, ready;

struct Parked { void* p; uint32_t size; void* who; };
static struct Parked* ring;
static uint32_t head, count;
static long long bytes;
static pthread_mutex_t lock = PTHREAD_MUTEX_INITIALIZER;
static long long parked_total, reports;
static int log_fd = -1;

static void resolve(void) {
    if (ready || resolving) return;
    resolving = 1;
    real_malloc = dlsym(RTLD_NEXT, "malloc");
    real_free = dlsym(RTLD_NEXT, "free");
    real_calloc = dlsym(RTLD_NEXT, "calloc");
    real_realloc = dlsym(RTLD_NEXT, "realloc");
    resolving = 0;
    ready = 1;
}

static void say(const char* s, size_t n) {
    (void)!write(2, s, n);
    if (log_fd < 0) {
        const char* base = getenv("HEAPGUARD_LOG");
        if (base) {
            char name[512];
            snprintf(name, sizeof name, "%s.%d", base, (int)getpid());
            log_fd = open(name, O_WRONLY | O_CREAT | O_APPEND, 0644);
        }
    }
    if (log_fd >= 0) (void)!write(log_fd, s, n);
}

static void check(const struct Parked* b, const char* when) {
    const unsigned char* q = (const unsigned char*)b->p;
    uint32_t i = 0;
    for (; i + 8 <= b->size; i += 8)
        if (*(const uint64_t*)(q + i) != 0xC5C5C5C5C5C5C5C5ull) break;
    for (; i < b->size; ++i)
        if (q[i] != FILL) break;
    if (i >= b->size) return;
    ++reports;
    char line[1024];
    Dl_info info;
    const char* lib = "?";
    const char* sym = "?";
    unsigned long off = 0, symoff = 0;
    if (b->who && dladdr(b->who, &info)) {
        if (info.dli_fname) lib = info.dli_fname;
        off = (unsigned long)((char*)b->who - (char*)info.dli_fbase);
        if (info.dli_sname) { sym = info.dli_sname; symoff = (unsigned long)((char*)b->who - (char*)info.dli_saddr); }
    }
    int n = snprintf(line, sizeof line, "HEAPGUARD write after free (%s): block %p of %u bytes, first changed byte at offset %u; freed by %s+0x%lx (%s+0x%lx)\n  bytes from offset %u:",
                     when, b->p, b->size, i, lib, off, sym, symoff, i & ~7u);
    for (uint32_t k = i & ~7u; k < b->size && k < (i & ~7u) + 32 && n < (int)sizeof line - 4; ++k)
        n += snprintf(line + n, sizeof line - n, " %02x", q[k]);
    n += snprintf(line + n, sizeof line - n, "\n");
    say(line, (size_t)n);
}

static void park(void* p, void* who) {
    if (!p) return;
    if ((char*)p >= boot && (char*)p < boot + sizeof boot) return;
    if (!ready) resolve();
    size_t size = malloc_usable_size(p);
    if (!ring || size < MIN_SIZE || size > MAX_SIZE) {
        real_free(p);
        return;
    }
    memset(p, FILL, size);
    struct Parked out[8];
    int nout = 0;
    pthread_mutex_lock(&lock);
    ring[(head + count) & (RING - 1)] = (struct Parked){p, (uint32_t)size, who};
    ++count;
    bytes += (long long)size;
    ++parked_total;
    while ((count >= RING || bytes > MAX_BYTES) && nout < 8) {
        out[nout] = ring[head];
        head = (head + 1) & (RING - 1);
        --count;
        bytes -= out[nout].size;
        ++nout;
    }
    pthread_mutex_unlock(&lock);
    for (int k = 0; k < nout; ++k) {
        check(&out[k], "leaving the quarantine");
        real_free(out[k].p);
    }
}

void* malloc(size_t n) {
    if (!ready) {
        resolve();
        if (!ready) {                                     /* dlsym's own allocations */
            size_t at = (boot_used + 15) & ~(size_t)15;
            if (at + n > sizeof boot) return NULL;
            boot_used = at + n;
            return boot + at;
        }
    }
    return real_malloc(n);
}

void* calloc(size_t a, size_t b) {
    if (!ready) {
        resolve();
        if (!ready) {
            void* p = malloc(a * b);
            if (p) memset(p, 0, a * b);
            return p;
        }
    }
    return real_calloc(a, b);
}

void* realloc(void* p, size_t n) {
    if (!ready) resolve();
    if (p && (char*)p >= boot && (char*)p < boot + sizeof boot) {
        void* q = real_malloc(n);
        if (q) memcpy(q, p, n < 4096 ? n : 4096);
        return q;
    }
    return real_realloc(p, n);
}

void free(void* p) { park(p, __builtin_return_address(0)); }
void cfree(void* p) { park(p, __builtin_return_address(0)); }
/* operator delete in its forms: (void*), (void*, size_t), [] of both, and the aligned ones */
void _ZdlPv(void* p) { park(p, __builtin_return_address(0)); }
void _ZdlPvm(void* p, size_t n) { (void)n; park(p, __builtin_return_address(0)); }
void _ZdaPv(void* p) { park(p, __builtin_return_address(0)); }
void _ZdaPvm(void* p, size_t n) { (void)n; park(p, __builtin_return_address(0)); }
void _ZdlPvSt11align_val_t(void* p, size_t a) { (void)a; park(p, __builtin_return_address(0)); }
void _ZdlPvmSt11align_val_t(void* p, size_t n, size_t a) { (void)n; (void)a; park(p, __builtin_return_address(0)); }
void _ZdaPvSt11align_val_t(void* p, size_t a) { (void)a; park(p, __builtin_return_address(0)); }
void _ZdaPvmSt11align_val_t(void* p, size_t n, size_t a) { (void)n; (void)a; park(p, __builtin_return_address(0)); }
void _ZdlPvRKSt9nothrow_t(void* p, const void* t) { (void)t; park(p, __builtin_return_address(0)); }
void _ZdaPvRKSt9nothrow_t(void* p, const void* t) { (void)t; park(p, __builtin_return_address(0)); }

__attribute__((constructor)) static void begin(void) {
    resolve();
    ring = real_calloc(RING, sizeof *ring);
}

__attribute__((destructor)) static void end(void) {
    pthread_mutex_lock(&lock);
    uint32_t n = count, h = head;
    pthread_mutex_unlock(&lock);
    for (uint32_t k = 0; k < n; ++k) check(&ring[(h + k) & (RING - 1)], "still parked at exit");
    char line[256];
    int m = snprintf(line, sizeof line, "HEAPGUARD pid %d: %lld blocks parked, %lld reports\n", (int)getpid(), parked_total, reports);
    say(line, (size_t)m);
}
