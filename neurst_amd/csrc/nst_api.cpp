// Error plumbing + ABI version for libneurst_hip.so.
#include <stdarg.h>

#include "nst_common.h"

static thread_local char g_err[512] = "";

void nst_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int nst_abi_version(void) { return NST_ABI_VERSION; }

// ---------------------------------------------------------------------------------------------
// dropout seed offset: one device scalar per process (one process drives one GPU)
// ---------------------------------------------------------------------------------------------
static uint64_t* g_seed_offset = nullptr;        // the library's own scalar (default)
static thread_local uint64_t* g_seed_bound = nullptr;   // the scalar the host bound for its next launches, if any

const uint64_t* nst_seed_offset_devptr() {
  if (g_seed_bound) return g_seed_bound;
  if (!g_seed_offset) {
    uint64_t* p = nullptr;
    if (hipMalloc((void**)&p, 64) != hipSuccess || hipMemset(p, 0, 64) != hipSuccess) {
      nst_set_error("dropout seed offset: device allocation failed");
      return nullptr;
    }
    g_seed_offset = p;
  }
  return g_seed_offset;
}

namespace {
__global__ void seed_offset_kernel(uint64_t* p, uint64_t v, int add) { *p = add ? *p + v : v; }
}  // namespace

static int seed_offset_update(uint64_t v, int add, void* stream) {
  uint64_t* p = const_cast<uint64_t*>(nst_seed_offset_devptr());
  if (!p) return NST_ERR_LAUNCH;
  seed_offset_kernel<<<1, 1, 0, (hipStream_t)stream>>>(p, v, add);
  NST_CHECK_LAUNCH("dropout_seed_offset");
  return NST_OK;
}
extern "C" int nst_dropout_seed_offset_bind(uint64_t* scalar_dev) {
  g_seed_bound = scalar_dev;
  return NST_OK;
}
extern "C" int nst_dropout_seed_offset_set(uint64_t value, void* stream) { return seed_offset_update(value, 0, stream); }
extern "C" int nst_dropout_seed_offset_add(uint64_t delta, void* stream) { return seed_offset_update(delta, 1, stream); }
extern "C" const char* nst_last_error_string(void) { return g_err; }

// ---------------------------------------------------------------------------------------------
// Streams of a given priority class.  The HIP runtime multiplexes the streams of ONE class onto a few hardware queues (least
// used first), and two busy streams that end up in one queue run in turns; the three concurrent activities of a training step
// -- the step itself, its weight-gradient stream, the gradient exchange -- are therefore kept in three different classes.
// ---------------------------------------------------------------------------------------------
extern "C" int nst_stream_create(int priority_class, void** stream_out) {
  NST_CHECK_ARG(stream_out != nullptr, "nst_stream_create: NULL output");
  *stream_out = nullptr;
  NST_CHECK_ARG(priority_class >= -1 && priority_class <= 1, "nst_stream_create: priority class %d (-1 high, 0 default, 1 low)",
                priority_class);
  int least = 0, greatest = 0;
  NST_CHECK_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
  const int prio = priority_class < 0 ? greatest : priority_class > 0 ? least : 0;
  if ((priority_class < 0 && greatest >= 0) || (priority_class > 0 && least <= 0)) {
    nst_set_error("nst_stream_create: the device has no priority class %d (range %d .. %d)", priority_class, greatest, least);
    return NST_ERR_UNSUPPORTED;
  }
  hipStream_t s = nullptr;
  NST_CHECK_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio));
  *stream_out = s;
  return NST_OK;
}
extern "C" int nst_stream_destroy(void* stream) {
  if (stream) NST_CHECK_HIP(hipStreamDestroy((hipStream_t)stream));
  return NST_OK;
}

// ---------------------------------------------------------------------------------------------
// Host helper of the data feed: CRC-32C (Castagnoli) of TFRecord frames, slicing-by-8 (the Python table walk in
// neurst_amd/data/tfrecord.py does ~2 MB/s; a 900-frame utterance is 288 KB).
// ---------------------------------------------------------------------------------------------
static uint32_t g_crc_tab[8][256];
static bool g_crc_ready = false;

static void crc32c_init() {
  for (int i = 0; i < 256; ++i) {
    uint32_t c = (uint32_t)i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
    g_crc_tab[0][i] = c;
  }
  for (int i = 0; i < 256; ++i)
    for (int t = 1; t < 8; ++t) g_crc_tab[t][i] = (g_crc_tab[t - 1][i] >> 8) ^ g_crc_tab[0][g_crc_tab[t - 1][i] & 0xFF];
  g_crc_ready = true;
}

extern "C" uint32_t nst_crc32c(const void* data, int64_t n, uint32_t crc) {
  if (!g_crc_ready) crc32c_init();  // idempotent; racing initialisers write the same values
  const unsigned char* p = (const unsigned char*)data;
  uint32_t c = crc ^ 0xFFFFFFFFu;
  while (n > 0 && ((uintptr_t)p & 7)) { c = g_crc_tab[0][(c ^ *p++) & 0xFF] ^ (c >> 8); --n; }
  while (n >= 8) {
    uint64_t w;
    memcpy(&w, p, 8);
    w ^= c;
    c = g_crc_tab[7][w & 0xFF] ^ g_crc_tab[6][(w >> 8) & 0xFF] ^ g_crc_tab[5][(w >> 16) & 0xFF] ^ g_crc_tab[4][(w >> 24) & 0xFF] ^
        g_crc_tab[3][(w >> 32) & 0xFF] ^ g_crc_tab[2][(w >> 40) & 0xFF] ^ g_crc_tab[1][(w >> 48) & 0xFF] ^ g_crc_tab[0][(w >> 56) & 0xFF];
    p += 8;
    n -= 8;
  }
  while (n-- > 0) c = g_crc_tab[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}
