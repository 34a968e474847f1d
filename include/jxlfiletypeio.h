/*
 * C-ABI of libJpegXLFileTypeIO (MI355X-native) — the drop-in boundary.
 *
 * Part 1 replaces, symbol for symbol, the three exports of the reference's native IO DLL:
 *   reference: src/JxlFileTypeIO/JxlFileTypeIO.h:29-43 (exports)
 *              src/JxlFileTypeIO/Common.h:17-60 (BitmapData, ImageChannelRepresentation, ProgressProc, IOCallbacks, ErrorInfo)
 *              src/JxlFileTypeIO/Decoder/JxlDecoderTypes.h:17-71 (DecoderStatus, DecoderImageFormat, KnownColorProfile, DecoderCallbacks)
 *              src/JxlFileTypeIO/Encoder/JxlEncoderTypes.h:17-42 (EncoderStatus, EncoderOptions, EncoderImageMetadata)
 *   bound by:  src/Interop/JpegXL_X64.cs:19-39 (LibraryImport "JpegXLFileTypeIO_X64.dll", stdcall)
 * `__stdcall` is a no-op on x64/ARM64 and is defined empty here.  Struct layouts are identical under LP64 and LLP64
 * (checked by the static asserts at the end).  `bool` is one byte (src/Interop/DecoderCallbacks.cs:22-34).
 *
 * Part 2 (jxlhip_*) is the device-resident batch entry point used by bench.py / tests: same decode path,
 * but bitstreams and RGBA8 outputs stay in HBM and several images are decoded per call.
 */
#ifndef JXLFILETYPEIO_H_
#define JXLFILETYPEIO_H_

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifndef JXL_STDCALL
#define JXL_STDCALL
#endif
#if defined(__GNUC__)
#define JXLFILETYPEIO_API __attribute__((visibility("default")))
#else
#define JXLFILETYPEIO_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Common.h:17-23 */
typedef struct BitmapData {
  uint8_t* scan0; /* BGRA8, `stride` bytes per row, host-owned, read-only */
  uint32_t width;
  uint32_t height;
  uint32_t stride;
} BitmapData;

/* ---- Common.h:33-39 */
typedef int32_t ImageChannelRepresentation;
enum { ImageChannelRepresentation_Uint8 = 0, ImageChannelRepresentation_Uint16, ImageChannelRepresentation_Float16, ImageChannelRepresentation_Float32 };

/* ---- Common.h:41-53 */
typedef bool(JXL_STDCALL* ProgressProc)(int32_t progressPercentage);
typedef int32_t(JXL_STDCALL* WriteCallback)(const uint8_t* buffer, size_t sizeInBytes); /* returns HRESULT */
typedef int32_t(JXL_STDCALL* SeekCallback)(uint64_t position);                          /* absolute; returns HRESULT */
typedef struct IOCallbacks {
  WriteCallback Write;
  SeekCallback Seek;
} IOCallbacks;

/* ---- Common.h:55-60 */
typedef struct ErrorInfo {
  char errorMessage[256]; /* NUL-terminated, written only when the message fits 255 chars (Common.cpp:18-53) */
} ErrorInfo;

/* ---- JxlDecoderTypes.h:17-32 */
typedef int32_t DecoderStatus;
enum {
  DecoderStatus_Ok = 0, DecoderStatus_NullParameter, DecoderStatus_InvalidParameter, DecoderStatus_OutOfMemory,
  DecoderStatus_HasAnimation, DecoderStatus_HasMultipleFrames, DecoderStatus_ImageDimensionExceedsInt32,
  DecoderStatus_UnsupportedChannelFormat, DecoderStatus_CreateLayerError, DecoderStatus_CreateMetadataError,
  DecoderStatus_DecodeError, DecoderStatus_MetadataError, DecoderStatus_InvalidFileSignature
};
/* ---- JxlDecoderTypes.h:34-39 */
typedef int32_t DecoderImageFormat;
enum { DecoderImageFormat_Gray = 0, DecoderImageFormat_Rgb, DecoderImageFormat_Cmyk };
/* ---- JxlDecoderTypes.h:41-51 */
typedef int32_t KnownColorProfile;
enum {
  KnownColorProfile_Srgb = 0, KnownColorProfile_LinearSrgb, KnownColorProfile_LinearGray, KnownColorProfile_GraySrgbTRC,
  KnownColorProfile_DisplayP3, KnownColorProfile_Rec709, KnownColorProfile_Rec2020Linear, KnownColorProfile_Rec2020PQ
};
/* ---- JxlDecoderTypes.h:53-71 */
typedef void(JXL_STDCALL* DecoderSetBasicInfo)(int32_t width, int32_t height, DecoderImageFormat format,
                                               ImageChannelRepresentation channelFormat, bool hasTransparency);
typedef bool(JXL_STDCALL* DecoderSetMetadata)(uint8_t* data, size_t length);
typedef bool(JXL_STDCALL* DecoderSetKnownColorProfile)(KnownColorProfile profile);
typedef bool(JXL_STDCALL* DecoderSetLayerData)(uint8_t* pixels, char* name, size_t nameLength);
typedef struct DecoderCallbacks {
  DecoderSetBasicInfo setBasicInfo;
  DecoderSetMetadata setIccProfile;
  DecoderSetKnownColorProfile setKnownColorProfile;
  DecoderSetMetadata setExif;
  DecoderSetMetadata setXmp;
  DecoderSetLayerData setLayerData;
} DecoderCallbacks;

/* ---- JxlEncoderTypes.h:17-25 */
typedef int32_t EncoderStatus;
enum { EncoderStatus_Ok = 0, EncoderStatus_NullParameter, EncoderStatus_OutOfMemory, EncoderStatus_UserCanceled,
       EncoderStatus_EncodeError, EncoderStatus_WriteError };
/* ---- JxlEncoderTypes.h:27-32 */
typedef struct EncoderOptions {
  float distance;
  int32_t effort;
  bool lossless;
} EncoderOptions;
/* ---- JxlEncoderTypes.h:34-42 */
typedef struct EncoderImageMetadata {
  uint8_t* exif; /* already carries the 4-byte big-endian TIFF offset prefix (src/Exif/ExifWriter.cs:86-90) */
  size_t exifSize;
  uint8_t* iccProfile;
  size_t iccProfileSize;
  uint8_t* xmp;
  size_t xmpSize;
} EncoderImageMetadata;

/* ---- JxlFileTypeIO.h:29 — (major<<24)|(minor<<16)|(patch<<8), unpacked by src/Interop/JpegXLNative.cs:40-42 */
JXLFILETYPEIO_API uint32_t JXL_STDCALL GetLibJxlVersion(void);

/* ---- JxlFileTypeIO.h:31-35 -> Decoder/JxlDecoder.cpp:796-852.
 * Callback order: setBasicInfo, then colour profile / Exif / XMP, then setLayerData exactly once.
 * Pixels handed to setLayerData: interleaved, tight rows, Gray|GrayA|RGB|RGBA, 8 bits per sample. */
JXLFILETYPEIO_API DecoderStatus JXL_STDCALL LoadImage(DecoderCallbacks* callbacks, const uint8_t* data, size_t dataSize,
                                                      ErrorInfo* errorInfo);

/* ---- JxlFileTypeIO.h:37-43 -> Encoder/JxlEncoder.cpp:147-392 */
JXLFILETYPEIO_API EncoderStatus JXL_STDCALL SaveImage(const BitmapData* bitmap, const EncoderOptions* options,
                                                      const EncoderImageMetadata* metadata, IOCallbacks* callbacks,
                                                      ErrorInfo* errorInfo, ProgressProc progressCallback);

/* =====================================================================================================
 * Part 2: device-resident batch decode (not in the reference; used by bench.py and the GPU parity tests).
 * ===================================================================================================== */
typedef struct JxlHipDecoder JxlHipDecoder;

typedef struct JxlHipImageInfo {
  int32_t width, height;
  int32_t num_channels;   /* 1..4, interleaved u8 */
  int32_t has_alpha;
  int32_t xsize_blocks, ysize_blocks;
  int32_t num_groups, num_lf_groups;
  int32_t epf_iters, gaborish;
  uint64_t codestream_bytes;
  int32_t bytes_per_sample;   /* 1: u8 output; 2: u16 (more than 8 bits per sample) or binary16; 4: binary32 */
  int32_t reserved;           /* 1: the samples are floats (binary16 / binary32) */
} JxlHipImageInfo;

/* device < 0: current HIP device.  Returns NULL on failure (message in err, may be NULL). */
JXLFILETYPEIO_API JxlHipDecoder* jxlhip_decoder_create(int32_t device, ErrorInfo* err);
JXLFILETYPEIO_API void jxlhip_decoder_destroy(JxlHipDecoder* dec);

/* Parses headers only (host). */
JXLFILETYPEIO_API DecoderStatus jxlhip_peek(const uint8_t* data, size_t size, JxlHipImageInfo* info, ErrorInfo* err);

/* Decodes n files.  host_data[i]/sizes[i]: the file bytes in host memory (headers are parsed on the host).
 * dev_data[i]: the same bytes already resident in HBM, or NULL (then they are uploaded inside the call).
 * dev_out[i]: device buffer of width*height*num_channels*bytes_per_sample bytes receiving interleaved u8 (or u16) pixels.
 * Work is enqueued on `stream` (a hipStream_t, may be NULL = default stream); the call returns after
 * enqueueing unless `synchronize` is non-zero.  Per-image status is written to statuses[i] on return when
 * synchronizing, otherwise by jxlhip_finish(). */
JXLFILETYPEIO_API DecoderStatus jxlhip_decode_batch(JxlHipDecoder* dec, int32_t n, const uint8_t* const* host_data,
                                                    const size_t* sizes, const uint8_t* const* dev_data, uint8_t* const* dev_out,
                                                    void* stream, int32_t synchronize, DecoderStatus* statuses, ErrorInfo* err);
/* Waits for the last batch and collects device-side error flags. */
JXLFILETYPEIO_API DecoderStatus jxlhip_finish(JxlHipDecoder* dec, DecoderStatus* statuses, ErrorInfo* err);

/* Stage taps of the most recent (synchronised) batch, image `index`, for parity tests.  `name` as in
 * DESIGN.md ("lf", "qcoef", "xyb_idct", "xyb_filtered", "strategy", "raw_quant", "sharpness", "alpha", ...).
 * Copies up to `capacity` bytes device->host; returns the full byte size of the plane (0 = unknown name). */
JXLFILETYPEIO_API size_t jxlhip_read_plane(JxlHipDecoder* dec, int32_t index, const char* name, int32_t channel, void* dst,
                                           size_t capacity);

/* Options: "debug_taps" (0/1: keep qcoef / xyb_idct / xyb_filtered stage copies; slow), "lane_stride" (0 = auto,
 * 64 = one section per wavefront ... 1 = one section per lane), "overlap" (0/1: run the LF stage of the next
 * asynchronous batch on a second stream while the previous batch finishes), "band_first_row" / "band_rows" (>= 0: decode only these
 * 256-pixel group rows of a lossy frame into a band-sized buffer), "no_direct" / "mod_lanes64" (launch shapes of the vector loops for
 * small launches as well; same output), "no_stream_pairs" (every fused Gaborish + EPF frame through the four-pixels-per-lane kernel;
 * same output).  Returns 1 if the option exists and the value is valid (0: refused, nothing changed). */
JXLFILETYPEIO_API int32_t jxlhip_set_option(JxlHipDecoder* dec, const char* name, int32_t value);

/* Timing of the last synchronised batch: milliseconds per named stage (HIP events on the decode stream). */
JXLFILETYPEIO_API int32_t jxlhip_stage_times(JxlHipDecoder* dec, const char** names, float* ms, int32_t capacity);

/* Cumulative per-stage time over every batch finished since the last reset (asynchronous batches included). */
JXLFILETYPEIO_API int32_t jxlhip_stage_totals(JxlHipDecoder* dec, const char** names, float* ms, int32_t capacity, int32_t* batches,
                                              int32_t reset);

/* Device time of the stages of the calling thread's last LoadImage / lossy SaveImage (HIP events on the streams the kernels were
 * launched on; measurement only: bench.py's single-image and encode workloads read them next to the wall time of the call). */
JXLFILETYPEIO_API int32_t jxlhip_last_load_stage_times(const char** names, float* ms, int32_t capacity);
JXLFILETYPEIO_API int32_t jxlhip_last_save_stage_times(const char** names, float* ms, int32_t capacity);

/* Host-only (no GPU): the embedded ICC profile as LoadImage would hand it to setIccProfile (reference Decoder/JxlDecoder.cpp:652-682);
 * returns its size (0: none) and copies up to `capacity` bytes. */
JXLFILETYPEIO_API size_t jxlhip_parse_icc(const uint8_t* data, size_t size, uint8_t* dst, size_t capacity, DecoderStatus* status, ErrorInfo* err);

#ifdef __cplusplus
}
static_assert(sizeof(BitmapData) == 24, "BitmapData layout");
static_assert(sizeof(EncoderOptions) == 12, "EncoderOptions layout");
static_assert(sizeof(EncoderImageMetadata) == 48, "EncoderImageMetadata layout");
static_assert(sizeof(IOCallbacks) == 16, "IOCallbacks layout");
static_assert(sizeof(ErrorInfo) == 256, "ErrorInfo layout");
static_assert(sizeof(DecoderCallbacks) == 48, "DecoderCallbacks layout");
static_assert(sizeof(bool) == 1, "bool is one byte across the boundary");
#endif

#endif /* JXLFILETYPEIO_H_ */
