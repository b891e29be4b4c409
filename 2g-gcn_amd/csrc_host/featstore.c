// Host-side reader of the reference's feature stores: Blosc-1 frame decoder (LZ4 / zlib inner codec, byte shuffle)
// and chunk-file reader that lands an array directly in the caller's (pinned) buffer. C ABI: include/twog_featstore.h.
//
// Formats restated here (the reference reaches them through zarr==2.4.0 -> numcodecs==0.6.4 -> bundled c-blosc):
//   * LZ4 block format: sequences of [token][literal length bytes][literals][offset le16][match length bytes]; the
//     last sequence stops after its literals; match length = low nibble + 4; offsets may overlap the output cursor.
//   * Blosc-1 frame: 16-byte header {version, versionlz, flags, typesize, nbytes, blocksize, cbytes}; flags bit 0 byte
//     shuffle, bit 1 memcpyed, bit 2 bit shuffle, bit 4 do-not-split, bits 5-7 inner codec (0 blosclz, 1 lz4/lz4hc,
//     2 snappy, 3 zlib, 4 zstd); then one le32 start offset per block; a block is `typesize` split streams (only when
//     splitting is on, typesize <= 16, blocksize / typesize >= 128 and it is not the short last block) or one stream,
//     each stream = le32 compressed length + payload, stored verbatim when that length equals the decoded length.
#include "../../include/twog_featstore.h"

#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#define BLOSC_HEADER 16
#define BLOSC_MAX_SPLITS 16
#define BLOSC_MIN_BUFFERSIZE 128
#define FLAG_SHUFFLE 0x1
#define FLAG_MEMCPYED 0x2
#define FLAG_BITSHUFFLE 0x4
#define FLAG_DONT_SPLIT 0x10

const char* twog_fs_version(void) { return "twog_featstore 1 (blosc-1 frames: lz4|zlib, byte shuffle; zarr v2 chunk files)"; }

static inline uint32_t le32(const uint8_t* p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

int64_t twog_lz4_block_decode(const uint8_t* src, int64_t src_len, uint8_t* dst, int64_t dst_cap) {
    if (!src || (!dst && dst_cap > 0) || src_len < 0 || dst_cap < 0) return TWOG_FS_EARG;
    const uint8_t *ip = src, *const iend = src + src_len;
    uint8_t *op = dst, *const oend = dst + dst_cap;
    for (;;) {
        if (ip >= iend) return TWOG_FS_ECORRUPT;
        const unsigned token = *ip++;
        size_t lit = token >> 4;
        if (lit == 15) {
            unsigned s;
            do {
                if (ip >= iend) return TWOG_FS_ECORRUPT;
                s = *ip++;
                lit += s;
            } while (s == 255);
        }
        if (lit <= 16 && (size_t)(iend - ip) >= 16 && (size_t)(oend - op) >= 16) {
            memcpy(op, ip, 16);  // short literal runs dominate: one fixed-size copy, the surplus is overwritten later
        } else {
            if ((size_t)(iend - ip) < lit) return TWOG_FS_ECORRUPT;
            if ((size_t)(oend - op) < lit) return TWOG_FS_ESPACE;
            memcpy(op, ip, lit);
        }
        ip += lit;
        op += lit;
        if (ip == iend) break;  // the last sequence carries literals only
        if (iend - ip < 2) return TWOG_FS_ECORRUPT;
        const size_t off = (size_t)ip[0] | ((size_t)ip[1] << 8);
        ip += 2;
        if (off == 0 || off > (size_t)(op - dst)) return TWOG_FS_ECORRUPT;
        size_t ml = token & 15;
        if (ml == 15) {
            unsigned s;
            do {
                if (ip >= iend) return TWOG_FS_ECORRUPT;
                s = *ip++;
                ml += s;
            } while (s == 255);
        }
        ml += 4;
        if ((size_t)(oend - op) < ml) return TWOG_FS_ESPACE;
        const uint8_t* m = op - off;
        if (off >= 8 && (size_t)(oend - op) >= ml + 8) {
            // 8-byte pieces never overlap their source (distance >= 8); up to 7 surplus bytes land inside the output
            // buffer and are overwritten by the next sequence
            uint8_t* const e = op + ml;
            do {
                memcpy(op, m, 8);
                op += 8;
                m += 8;
            } while (op < e);
            op = e;
            continue;
        }
        if (off < 8 && ml >= 32) {
            // short-period run (e.g. off = 1: one repeated byte): lay down the period byte-wise until the distance
            // between read and write cursor is >= 8, then continue in 8-byte pieces (a piece never overlaps its source)
            size_t dist = off;
            while (dist < 8) {
                for (size_t k = 0; k < dist && ml; ++k, --ml) *op++ = *m++;
                m = op - 2 * dist;  // the run so far is periodic in `off`, hence also in every multiple of it
                dist *= 2;
            }
            while (ml >= 8) {
                memcpy(op, m, 8);
                op += 8;
                m += 8;
                ml -= 8;
            }
        }
        while (ml--) *op++ = *m++;  // overlapping tail / short-period runs: byte order matters
    }
    return (int64_t)(op - dst);
}

int twog_blosc_info(const uint8_t* frame, int64_t frame_len, twog_blosc_info_t* info) {
    if (!frame || !info || frame_len < 0) return TWOG_FS_EARG;
    if (frame_len < BLOSC_HEADER) return TWOG_FS_EHEADER;
    info->version = frame[0];
    info->versionlz = frame[1];
    info->flags = frame[2];
    info->typesize = frame[3];
    info->nbytes = le32(frame + 4);
    info->blocksize = le32(frame + 8);
    info->cbytes = le32(frame + 12);
    if (info->version != 2) return TWOG_FS_EHEADER;  // BLOSC_VERSION_FORMAT of every c-blosc 1.x release
    if (info->typesize < 1) return TWOG_FS_EHEADER;
    if (info->nbytes > 0x7fffffff || info->blocksize > 0x7fffffff || info->cbytes > 0x7fffffff) return TWOG_FS_EHEADER;
    if (info->cbytes < BLOSC_HEADER || info->cbytes > frame_len) return TWOG_FS_EHEADER;
    if (info->nbytes > 0 && (info->blocksize <= 0 || info->blocksize > info->nbytes)) return TWOG_FS_EHEADER;
    return 0;
}

// shuffled block: byte j of every element first (ne = bsize / ts elements), the bytes past the last whole element last
static void unshuffle(int ts, int64_t bsize, const uint8_t* src, uint8_t* dst) {
    const int64_t ne = bsize / ts;
    if (ts == 4) {
        const uint8_t *s0 = src, *s1 = src + ne, *s2 = src + 2 * ne, *s3 = src + 3 * ne;
        int64_t i = 0;
#if defined(__SSE2__)
        // 16 elements per step: two rounds of byte / 16-bit interleaves turn four byte planes into 64 output bytes
        for (; i + 16 <= ne; i += 16) {
            const __m128i p0 = _mm_loadu_si128((const __m128i*)(s0 + i)), p1 = _mm_loadu_si128((const __m128i*)(s1 + i));
            const __m128i p2 = _mm_loadu_si128((const __m128i*)(s2 + i)), p3 = _mm_loadu_si128((const __m128i*)(s3 + i));
            const __m128i a = _mm_unpacklo_epi8(p0, p1), b = _mm_unpackhi_epi8(p0, p1);
            const __m128i c = _mm_unpacklo_epi8(p2, p3), d = _mm_unpackhi_epi8(p2, p3);
            __m128i* o = (__m128i*)(dst + 4 * i);
            _mm_storeu_si128(o + 0, _mm_unpacklo_epi16(a, c));
            _mm_storeu_si128(o + 1, _mm_unpackhi_epi16(a, c));
            _mm_storeu_si128(o + 2, _mm_unpacklo_epi16(b, d));
            _mm_storeu_si128(o + 3, _mm_unpackhi_epi16(b, d));
        }
#endif
        for (; i < ne; ++i) {
            const uint32_t v = (uint32_t)s0[i] | ((uint32_t)s1[i] << 8) | ((uint32_t)s2[i] << 16) | ((uint32_t)s3[i] << 24);
            memcpy(dst + 4 * i, &v, 4);  // host is little-endian (x86-64); byte order of the store is kept as is
        }
    } else {
        for (int j = 0; j < ts; ++j) {
            const uint8_t* s = src + (int64_t)j * ne;
            for (int64_t i = 0; i < ne; ++i) dst[i * ts + j] = s[i];
        }
    }
    const int64_t done = ne * ts;
    if (bsize > done) memcpy(dst + done, src + done, (size_t)(bsize - done));
}

typedef struct {
    const uint8_t* frame;
    int64_t cbytes;
    uint8_t* dst;
    int64_t nbytes, blocksize;
    int typesize, flags, codec;
    int64_t nblocks;
    int64_t next;  // work counter (atomic)
    int err;       // first error (atomic)
} job_t;

static int decode_block(const job_t* J, int64_t j, uint8_t* tmp) {
    const int64_t leftover = J->nbytes % J->blocksize;
    const int last_short = (j == J->nblocks - 1) && leftover > 0;
    const int64_t bsize = last_short ? leftover : J->blocksize;
    const int64_t table_end = BLOSC_HEADER + 4 * J->nblocks;
    int64_t pos = (int64_t)le32(J->frame + BLOSC_HEADER + 4 * j);
    if (pos < table_end || pos > J->cbytes) return TWOG_FS_ECORRUPT;
    const int ts = J->typesize;
    const int split = !(J->flags & FLAG_DONT_SPLIT) && ts <= BLOSC_MAX_SPLITS && (bsize / ts) >= BLOSC_MIN_BUFFERSIZE &&
                      !last_short;
    const int nsplits = split ? ts : 1;
    const int64_t neblock = bsize / nsplits;
    const int shuffled = (J->flags & FLAG_SHUFFLE) && ts > 1;
    uint8_t* const out_block = J->dst + j * J->blocksize;
    uint8_t* out = shuffled ? tmp : out_block;
    for (int s = 0; s < nsplits; ++s) {
        if (J->cbytes - pos < 4) return TWOG_FS_ECORRUPT;
        const int64_t clen = (int32_t)le32(J->frame + pos);
        pos += 4;
        if (clen < 0 || clen > J->cbytes - pos) return TWOG_FS_ECORRUPT;
        if (clen == neblock) {
            memcpy(out, J->frame + pos, (size_t)neblock);  // stored stream
        } else if (J->codec == 1) {
            const int64_t got = twog_lz4_block_decode(J->frame + pos, clen, out, neblock);
            if (got != neblock) return TWOG_FS_ECORRUPT;
        } else {
            uLongf dlen = (uLongf)neblock;
            if (uncompress(out, &dlen, J->frame + pos, (uLong)clen) != Z_OK || (int64_t)dlen != neblock)
                return TWOG_FS_ECORRUPT;
        }
        pos += clen;
        out += neblock;
    }
    if (shuffled) unshuffle(ts, bsize, tmp, out_block);
    return 0;
}

static void* worker(void* arg) {
    job_t* J = (job_t*)arg;
    uint8_t* tmp = (uint8_t*)malloc((size_t)J->blocksize);
    if (!tmp) {
        int expect = 0;
        __atomic_compare_exchange_n(&J->err, &expect, TWOG_FS_ENOMEM, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
        return NULL;
    }
    for (;;) {
        const int64_t j = __atomic_fetch_add(&J->next, 1, __ATOMIC_RELAXED);
        if (j >= J->nblocks || __atomic_load_n(&J->err, __ATOMIC_RELAXED)) break;
        const int rc = decode_block(J, j, tmp);
        if (rc) {
            int expect = 0;
            __atomic_compare_exchange_n(&J->err, &expect, rc, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
            break;
        }
    }
    free(tmp);
    return NULL;
}

int64_t twog_blosc_decode(const uint8_t* frame, int64_t frame_len, uint8_t* dst, int64_t dst_cap, int n_threads) {
    twog_blosc_info_t h;
    if (!frame || (!dst && dst_cap > 0) || dst_cap < 0) return TWOG_FS_EARG;
    const int rc = twog_blosc_info(frame, frame_len, &h);
    if (rc) return rc;
    if (h.nbytes > dst_cap) return TWOG_FS_ESPACE;
    if (h.nbytes == 0) return 0;
    if (h.flags & FLAG_MEMCPYED) {
        if (BLOSC_HEADER + h.nbytes > h.cbytes) return TWOG_FS_ECORRUPT;
        memcpy(dst, frame + BLOSC_HEADER, (size_t)h.nbytes);
        return h.nbytes;
    }
    if (h.flags & FLAG_BITSHUFFLE) return TWOG_FS_ECODEC;
    const int codec = (h.flags >> 5) & 7;
    if (codec != 1 && codec != 3) return TWOG_FS_ECODEC;
    job_t J;
    J.frame = frame;
    J.cbytes = h.cbytes;
    J.dst = dst;
    J.nbytes = h.nbytes;
    J.blocksize = h.blocksize;
    J.typesize = h.typesize;
    J.flags = h.flags;
    J.codec = codec;
    J.nblocks = (h.nbytes + h.blocksize - 1) / h.blocksize;
    J.next = 0;
    J.err = 0;
    if (BLOSC_HEADER + 4 * J.nblocks > h.cbytes) return TWOG_FS_ECORRUPT;
    int nt = n_threads;
    if (nt > J.nblocks) nt = (int)J.nblocks;
    if (nt > 64) nt = 64;
    if (h.nbytes < (256 << 10)) nt = 1;  // thread start-up costs more than a small frame
    if (nt <= 1) {
        worker(&J);
    } else {
        pthread_t th[64];
        int started = 0;
        for (int i = 0; i < nt - 1; ++i) {
            if (pthread_create(&th[started], NULL, worker, &J) != 0) break;
            ++started;
        }
        worker(&J);  // the caller takes a share (and all of it if no thread could be started)
        for (int i = 0; i < started; ++i) pthread_join(th[i], NULL);
    }
    if (J.err) return J.err;
    return h.nbytes;
}

static int read_all(int fd, uint8_t* dst, int64_t n) {
    int64_t got = 0;
    while (got < n) {
        const ssize_t r = read(fd, dst + got, (size_t)(n - got));
        if (r < 0) {
            if (errno == EINTR) continue;
            return TWOG_FS_EIO;
        }
        if (r == 0) return TWOG_FS_EIO;
        got += r;
    }
    return 0;
}

int64_t twog_fs_read_chunk(const char* path, int codec, uint8_t* dst, int64_t nbytes, int n_threads) {
    if (!path || (!dst && nbytes > 0) || nbytes < 0) return TWOG_FS_EARG;
    if (codec != 0 && codec != 1) return TWOG_FS_ECODEC;
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return TWOG_FS_EIO;
    struct stat st;
    if (fstat(fd, &st) != 0) {
        close(fd);
        return TWOG_FS_EIO;
    }
    int64_t rc;
    if (codec == 0) {
        if ((int64_t)st.st_size != nbytes) {
            rc = TWOG_FS_ESIZE;
        } else {
            rc = read_all(fd, dst, nbytes);  // straight into the caller's (pinned) buffer
            if (rc == 0) rc = nbytes;
        }
    } else {
        uint8_t* frame = (uint8_t*)malloc(st.st_size > 0 ? (size_t)st.st_size : 1);
        if (!frame) {
            rc = TWOG_FS_ENOMEM;
        } else {
            rc = read_all(fd, frame, (int64_t)st.st_size);
            if (rc == 0) {
                twog_blosc_info_t h;
                rc = twog_blosc_info(frame, (int64_t)st.st_size, &h);
                if (rc == 0) rc = (h.nbytes != nbytes) ? TWOG_FS_ESIZE : twog_blosc_decode(frame, (int64_t)st.st_size, dst, nbytes, n_threads);
            }
            free(frame);
        }
    }
    close(fd);
    return rc;
}
