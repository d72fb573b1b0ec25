// FLAC stream decoder for the input side of the `enhance` CLI (pure host C++, no HIP).
// The reference reads its input files with torchaudio.load (open_universe/bin/enhance.py:183; AUDIO_SUFFIXES :33 lists .flac),
// whose FLAC codec is native code; torchaudio is absent from this image, speech corpora ship as FLAC (LibriSpeech, VCTK 0.92).
// Restated from the published format (xiph.org FLAC format specification / RFC 9639), not from any source file: STREAMINFO,
// frame headers (fixed / variable block size, every block-size / sample-rate / sample-size code, CRC-8), the four subframe types
// (constant, verbatim, fixed predictors of order 0-4, LPC of order 1-32 with wasted bits), partitioned Rice residuals (4- and
// 5-bit parameters, escaped partitions), the three stereo decorrelation modes, frame CRC-16.  Every checksum is verified: a
// stream this decoder misreads is refused, not passed on; the MD5 of the decoded audio is checked by the caller (audio.py).
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ouniverse.h"

namespace {

thread_local std::string g_flac_error;

int flac_fail(const std::string& m) {
  g_flac_error = m;
  return OU_EINVAL;
}

struct Bits {
  const uint8_t* p;
  size_t n, pos = 0;  // bit position
  bool bad = false;
  Bits(const uint8_t* p_, size_t n_) : p(p_), n(n_) {}
  uint32_t get(int k) {  // k <= 32, MSB first
    uint64_t v = 0;
    for (int got = 0; got < k;) {
      const size_t byte = pos >> 3;
      if (byte >= n) { bad = true; return 0; }
      const int avail = 8 - (int)(pos & 7), take = k - got < avail ? k - got : avail;
      v = (v << take) | ((p[byte] >> (avail - take)) & ((1u << take) - 1));
      pos += take; got += take;
    }
    return (uint32_t)v;
  }
  int64_t sget(int k) {  // signed two's complement, k <= 33
    if (k == 0) return 0;
    uint64_t v = k > 32 ? ((uint64_t)get(k - 32) << 32) | get(32) : get(k);
    const uint64_t sign = 1ull << (k - 1);
    return (int64_t)((v ^ sign) - sign);
  }
  uint32_t unary() {  // number of 0 bits in front of the next 1 bit
    uint32_t q = 0;
    while (true) {
      const size_t byte = pos >> 3;
      if (byte >= n) { bad = true; return q; }
      const int off = (int)(pos & 7);
      const uint8_t rest = (uint8_t)(p[byte] << off);
      if (rest) {
        const int lz = __builtin_clz((unsigned)rest) - 24;
        q += lz; pos += lz + 1;
        return q;
      }
      q += 8 - off; pos += 8 - off;
    }
  }
  void align() { pos = (pos + 7) & ~(size_t)7; }
};

uint8_t crc8(const uint8_t* p, size_t n) {
  uint8_t c = 0;
  for (size_t i = 0; i < n; i++) {
    c ^= p[i];
    for (int b = 0; b < 8; b++) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : (c << 1));
  }
  return c;
}
uint16_t crc16(const uint8_t* p, size_t n) {
  uint16_t c = 0;
  for (size_t i = 0; i < n; i++) {
    c ^= (uint16_t)(p[i] << 8);
    for (int b = 0; b < 8; b++) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : (c << 1));
  }
  return c;
}

struct Info {
  int fs = 0, ch = 0, bps = 0, min_block = 0, max_block = 0;
  int64_t total = 0;
  uint8_t md5[16] = {0};
  size_t audio_off = 0;  // first frame
};

int parse_header(const uint8_t* d, size_t n, Info& I) {
  size_t off = 0;
  if (n >= 10 && d[0] == 'I' && d[1] == 'D' && d[2] == '3') {  // ID3v2 tag in front of the stream: sync-safe size
    const size_t sz = ((size_t)(d[6] & 0x7F) << 21) | ((size_t)(d[7] & 0x7F) << 14) | ((size_t)(d[8] & 0x7F) << 7) | (d[9] & 0x7F);
    off = 10 + sz + ((d[5] & 0x10) ? 10 : 0);
  }
  if (off + 4 > n || std::memcmp(d + off, "fLaC", 4) != 0) return flac_fail("not a FLAC stream (no fLaC marker)");
  off += 4;
  bool have = false;
  while (true) {
    if (off + 4 > n) return flac_fail("FLAC: truncated metadata");
    const bool last = (d[off] & 0x80) != 0;
    const int type = d[off] & 0x7F;
    const size_t len = ((size_t)d[off + 1] << 16) | ((size_t)d[off + 2] << 8) | d[off + 3];
    off += 4;
    if (off + len > n) return flac_fail("FLAC: truncated metadata block");
    if (type == 0) {
      if (len < 34) return flac_fail("FLAC: short STREAMINFO");
      const uint8_t* s = d + off;
      I.min_block = (s[0] << 8) | s[1];
      I.max_block = (s[2] << 8) | s[3];
      I.fs = (s[10] << 12) | (s[11] << 4) | (s[12] >> 4);
      I.ch = ((s[12] >> 1) & 7) + 1;
      I.bps = (((s[12] & 1) << 4) | (s[13] >> 4)) + 1;
      I.total = ((int64_t)(s[13] & 0xF) << 32) | ((int64_t)s[14] << 24) | ((int64_t)s[15] << 16) | ((int64_t)s[16] << 8) | s[17];
      std::memcpy(I.md5, s + 18, 16);
      have = true;
    }
    off += len;
    if (last) break;
  }
  if (!have) return flac_fail("FLAC: no STREAMINFO block");
  if (I.fs <= 0 || I.bps < 4 || I.bps > 32) return flac_fail("FLAC: invalid STREAMINFO");
  I.audio_off = off;
  return OU_OK;
}

// residual of one subframe into r[order .. bs)
bool read_residual(Bits& b, int64_t* r, int bs, int order) {
  const int method = (int)b.get(2);
  if (method > 1) return false;
  const int pbits = method == 0 ? 4 : 5, esc = method == 0 ? 15 : 31;
  const int po = (int)b.get(4), parts = 1 << po;
  if ((bs >> po) << po != bs && po > 0) return false;
  int i = order;
  for (int pt = 0; pt < parts; pt++) {
    int cnt = (bs >> po) - (pt == 0 ? order : 0);
    if (po == 0) cnt = bs - order;
    if (cnt < 0) return false;
    const int k = (int)b.get(pbits);
    if (k == esc) {
      const int nb = (int)b.get(5);
      for (int j = 0; j < cnt; j++) r[i++] = b.sget(nb);
    } else {
      for (int j = 0; j < cnt; j++) {
        const uint32_t q = b.unary();
        const uint64_t u = ((uint64_t)q << k) | (k ? b.get(k) : 0u);
        r[i++] = (int64_t)(u >> 1) ^ -(int64_t)(u & 1);
      }
    }
    if (b.bad) return false;
  }
  return i == bs;
}

bool read_subframe(Bits& b, int64_t* s, int bs, int bps) {
  if (b.get(1)) return false;
  const int type = (int)b.get(6);
  int wasted = 0;
  if (b.get(1)) wasted = (int)b.unary() + 1;
  bps -= wasted;
  if (bps <= 0 || b.bad) return false;
  if (type == 0) {
    const int64_t v = b.sget(bps);
    for (int i = 0; i < bs; i++) s[i] = v;
  } else if (type == 1) {
    for (int i = 0; i < bs; i++) s[i] = b.sget(bps);
  } else if (type >= 8 && type <= 12) {
    const int order = type - 8;
    if (order > bs) return false;
    for (int i = 0; i < order; i++) s[i] = b.sget(bps);
    if (!read_residual(b, s, bs, order)) return false;
    for (int i = order; i < bs; i++) {
      switch (order) {
        case 0: break;
        case 1: s[i] += s[i - 1]; break;
        case 2: s[i] += 2 * s[i - 1] - s[i - 2]; break;
        case 3: s[i] += 3 * s[i - 1] - 3 * s[i - 2] + s[i - 3]; break;
        default: s[i] += 4 * s[i - 1] - 6 * s[i - 2] + 4 * s[i - 3] - s[i - 4]; break;
      }
    }
  } else if (type >= 32) {
    const int order = type - 31;
    if (order > bs) return false;
    for (int i = 0; i < order; i++) s[i] = b.sget(bps);
    const int prec = (int)b.get(4) + 1;
    if (prec == 16) return false;
    const int shift = (int)b.sget(5);
    if (shift < 0) return false;
    int64_t c[32];
    for (int j = 0; j < order; j++) c[j] = b.sget(prec);
    if (!read_residual(b, s, bs, order)) return false;
    for (int i = order; i < bs; i++) {
      int64_t acc = 0;
      for (int j = 0; j < order; j++) acc += c[j] * s[i - 1 - j];
      s[i] += acc >> shift;
    }
  } else {
    return false;  // reserved subframe type
  }
  if (wasted)
    for (int i = 0; i < bs; i++) s[i] = (int64_t)((uint64_t)s[i] << wasted);
  return !b.bad;
}

// out == nullptr: count only.  out: [ch][cap] int32.
int decode(const uint8_t* d, size_t n, const Info& I, int32_t* out, int64_t cap, int64_t* decoded) {
  size_t off = I.audio_off;
  int64_t done = 0;
  std::vector<int64_t> buf;
  static const int kBlock[16] = {0, 192, 576, 1152, 2304, 4608, -8, -16, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768};
  static const int kBits[8] = {0, 8, 12, -1, 16, 20, 24, 32};
  while (off + 2 <= n) {
    if (!(d[off] == 0xFF && (d[off + 1] & 0xFE) == 0xF8)) {
      // trailing data behind the last frame (ID3v1 tags, padding) is tolerated once the announced samples are there
      if (I.total && done >= I.total) break;
      if (!I.total && done > 0) break;
      return flac_fail("FLAC: lost frame sync at byte " + std::to_string(off));
    }
    Bits b(d + off, n - off);
    b.get(15);
    b.get(1);  // blocking strategy: only changes the meaning of the coded number
    const int bsc = (int)b.get(4), src = (int)b.get(4), chc = (int)b.get(4), ssc = (int)b.get(3);
    if (b.get(1)) return flac_fail("FLAC: reserved frame-header bit set");
    {  // UTF-8-style coded frame / sample number
      const uint32_t first = b.get(8);
      int extra = 0;
      if (first >= 0xFE) extra = 6; else if (first >= 0xFC) extra = 5; else if (first >= 0xF8) extra = 4;
      else if (first >= 0xF0) extra = 3; else if (first >= 0xE0) extra = 2; else if (first >= 0xC0) extra = 1;
      else if (first >= 0x80) return flac_fail("FLAC: invalid coded frame number");
      for (int i = 0; i < extra; i++)
        if ((b.get(8) & 0xC0) != 0x80) return flac_fail("FLAC: invalid coded frame number");
    }
    int bs = kBlock[bsc];
    if (bs == 0) return flac_fail("FLAC: reserved block-size code");
    if (bs == -8) bs = (int)b.get(8) + 1;
    else if (bs == -16) bs = (int)b.get(16) + 1;
    if (src == 12) b.get(8);
    else if (src == 13 || src == 14) b.get(16);
    else if (src == 15) return flac_fail("FLAC: invalid sample-rate code");
    int bps = kBits[ssc];
    if (bps < 0) return flac_fail("FLAC: reserved sample-size code");
    if (bps == 0) bps = I.bps;
    // (a frame may name its own sample size; the caller scales by the stream-level one, so a frame that disagrees with
    //  STREAMINFO would come out at the wrong level: refused -- RFC 9639 streams keep one size throughout)
    if (bps != I.bps) return flac_fail("FLAC: frame sample size differs from STREAMINFO");
    const size_t hdr_bytes = b.pos >> 3;
    const uint8_t c8 = (uint8_t)b.get(8);
    if (b.bad) return flac_fail("FLAC: truncated frame header");
    if (crc8(d + off, hdr_bytes) != c8) return flac_fail("FLAC: frame-header CRC-8 mismatch at byte " + std::to_string(off));
    int nch;
    if (chc < 8) nch = chc + 1;
    else if (chc <= 10) nch = 2;
    else return flac_fail("FLAC: reserved channel assignment");
    if (nch != I.ch) return flac_fail("FLAC: channel count changes inside the stream");
    buf.assign((size_t)nch * bs, 0);
    for (int c = 0; c < nch; c++) {
      const bool side = (chc == 8 && c == 1) || (chc == 9 && c == 0) || (chc == 10 && c == 1);
      if (!read_subframe(b, buf.data() + (size_t)c * bs, bs, bps + (side ? 1 : 0)))
        return flac_fail("FLAC: invalid subframe in the frame at byte " + std::to_string(off));
    }
    b.align();
    const size_t body = b.pos >> 3;
    const uint16_t c16 = (uint16_t)b.get(16);
    if (b.bad) return flac_fail("FLAC: truncated frame");
    if (crc16(d + off, body) != c16) return flac_fail("FLAC: frame CRC-16 mismatch at byte " + std::to_string(off));
    int64_t* c0 = buf.data();
    int64_t* c1 = buf.data() + bs;
    if (chc == 8) {
      for (int i = 0; i < bs; i++) c1[i] = c0[i] - c1[i];
    } else if (chc == 9) {
      for (int i = 0; i < bs; i++) c0[i] = c0[i] + c1[i];
    } else if (chc == 10) {
      for (int i = 0; i < bs; i++) {
        const int64_t side = c1[i], mid = (int64_t)((uint64_t)c0[i] << 1) | (side & 1);
        c0[i] = (mid + side) >> 1;
        c1[i] = (mid - side) >> 1;
      }
    }
    int take = bs;
    if (I.total && done + take > I.total) take = (int)(I.total - done);
    if (out) {
      if (done + take > cap) return flac_fail("FLAC: output buffer too small");
      for (int c = 0; c < nch; c++)
        for (int i = 0; i < take; i++) out[(size_t)c * cap + done + i] = (int32_t)buf[(size_t)c * bs + i];
    }
    done += take;
    off += b.pos >> 3;
  }
  if (I.total && done != I.total) return flac_fail("FLAC: stream ends after " + std::to_string(done) + " of " + std::to_string(I.total) + " samples");
  *decoded = done;
  return OU_OK;
}

}  // namespace

extern "C" {

const char* ou_flac_last_error(void) { return g_flac_error.c_str(); }

int ou_flac_info(const uint8_t* data, size_t bytes, int32_t* sample_rate, int32_t* channels, int32_t* bits_per_sample,
                 int64_t* total_samples, uint8_t* md5) {
  if (!data) return flac_fail("bad argument");
  Info I;
  const int rc = parse_header(data, bytes, I);
  if (rc != OU_OK) return rc;
  if (sample_rate) *sample_rate = I.fs;
  if (channels) *channels = I.ch;
  if (bits_per_sample) *bits_per_sample = I.bps;
  if (md5) std::memcpy(md5, I.md5, 16);
  if (total_samples) {
    *total_samples = I.total;
    if (!I.total) {  // unknown in the header: count
      int64_t cnt = 0;
      const int rc2 = decode(data, bytes, I, nullptr, 0, &cnt);
      if (rc2 != OU_OK) return rc2;
      *total_samples = cnt;
    }
  }
  return OU_OK;
}

int ou_flac_decode(const uint8_t* data, size_t bytes, int32_t* out, int64_t capacity_per_channel, int64_t* decoded) {
  if (!data || !out || !decoded) return flac_fail("bad argument");
  Info I;
  const int rc = parse_header(data, bytes, I);
  if (rc != OU_OK) return rc;
  return decode(data, bytes, I, out, capacity_per_channel, decoded);
}

}  // extern "C"
