/*
 * twog_featstore.h -- C ABI of libtwog_featstore.so: host-side reader of the reference's on-disk feature format.
 *
 * The reference keeps the per-video Faster-RCNN features, bounding boxes and poses in zarr v2 directory stores and
 * reads them with `zarr.open(path, mode='r')[video_id][name][:]` (vhoi/data_loading.py:28,39-42,71-87,123-141); the
 * stores are written by `group.array(name, data, chunks=False, dtype=np.float32)` (vhoi/roi_features.py:227-242,
 * 292-295) with zarr's default compressor, i.e. ONE chunk file per array holding a Blosc-1 frame (cname lz4, clevel 5,
 * byte shuffle, itemsize 4; zarr==2.4.0 / numcodecs==0.6.4, reference environment.yml:113,125 -- both absent from
 * this image, so the container format below is restated from the published Blosc-1 chunk format and pinned against
 * frames produced by the real c-blosc 1.21.0, see tests/golden/g9_featstore and tools/make_golden_featstore.py).
 *
 * This library is plain C (no HIP): the work is byte shuffling and LZ4 match copies on the host. Its job on an MI355X
 * node is to land every array directly in the caller's (pinned) staging buffer, from which the DevicePrefetcher issues
 * one asynchronous H2D copy per batch -- no numpy temporaries, no second copy.
 *
 * Conventions: all pointers are HOST pointers owned by the caller; nothing is retained after return; every entry
 * point is re-entrant; return value >= 0 on success, a negative TWOG_FS_E* code otherwise; malformed or truncated
 * input is rejected with an error, never read or written out of bounds.
 */
#ifndef TWOG_FEATSTORE_H
#define TWOG_FEATSTORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    TWOG_FS_EARG = -1,      /* NULL pointer / negative size                                         */
    TWOG_FS_EHEADER = -2,   /* not a Blosc-1 frame (short, unknown version, inconsistent sizes)     */
    TWOG_FS_ESPACE = -3,    /* destination smaller than the decoded size                            */
    TWOG_FS_ECORRUPT = -4,  /* block table / stream lengths / LZ4 sequence out of bounds            */
    TWOG_FS_ECODEC = -5,    /* inner codec or filter this reader does not implement                 */
    TWOG_FS_EIO = -6,       /* open/read failed                                                     */
    TWOG_FS_ESIZE = -7,     /* chunk decodes to a size different from the one the array metadata asks */
    TWOG_FS_ENOMEM = -8
};

/* Human-readable build string. */
const char* twog_fs_version(void);

/* LZ4 block format decoder (the inner codec of the frames; LZ4 block format description, lz4 v1.x). Decodes exactly
 * one block of `src_len` bytes into at most `dst_cap` bytes; returns the number of bytes produced. */
int64_t twog_lz4_block_decode(const uint8_t* src, int64_t src_len, uint8_t* dst, int64_t dst_cap);

/* Fields of the 16-byte Blosc-1 header (c-blosc README_CHUNK_FORMAT: version, versionlz, flags, typesize, nbytes,
 * blocksize, cbytes). Returns 0 when `frame` starts with a well-formed header. */
typedef struct {
    int32_t version, versionlz, flags, typesize;
    int64_t nbytes;    /* decoded size            */
    int64_t blocksize; /* decoded bytes per block */
    int64_t cbytes;    /* size of the whole frame */
} twog_blosc_info_t;
int twog_blosc_info(const uint8_t* frame, int64_t frame_len, twog_blosc_info_t* info);

/* Decodes one Blosc-1 frame (what numcodecs.Blosc.decode does for a zarr chunk): per block, the split streams are
 * LZ4-decoded (or copied when stored), then byte-unshuffled into place. Implemented: memcpyed frames, inner codecs
 * lz4/lz4hc and zlib, byte shuffle or none, split and unsplit blocks; bit-shuffle and the other inner codecs return
 * TWOG_FS_ECODEC. Blocks are independent and are decoded by up to `n_threads` host threads (<= 1: the calling thread).
 * Returns the number of bytes written (== header nbytes). */
int64_t twog_blosc_decode(const uint8_t* frame, int64_t frame_len, uint8_t* dst, int64_t dst_cap, int n_threads);

/* Reads one chunk file of a zarr v2 directory store into `dst`, which must hold exactly `nbytes` (the chunk's decoded
 * size from the array metadata). codec: 0 = no compressor (the file is read straight into dst), 1 = Blosc frame.
 * Replaces zarr.core.Array._chunk_getitem for the single-chunk arrays the reference writes. Returns nbytes. */
int64_t twog_fs_read_chunk(const char* path, int codec, uint8_t* dst, int64_t nbytes, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
