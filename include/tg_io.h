/* tg_io.h — host-side input pipeline of libtg_hip.so: the on-disk format the reference's
 * Input_Pipeline/{cifar10,svhn,mnist}Dataset.py read through TensorFlow.
 *
 *   tf.data.TFRecordDataset(name)        (cifar10Dataset.py:38)     -> tg_ds_open / tg_ds_size / tg_ds_record
 *   tf.parse_single_example + decode_raw (cifar10Dataset.py:41-56)  -> tg_example_parse / tg_ds_gather
 *   (writing the files: not in the reference repository)            -> tg_tfrecord_write
 *
 * A TFRecord file is a sequence of { uint64 len | uint32 masked_crc32c(len) | payload[len] | uint32 masked_crc32c(payload) }
 * (little endian; CRC-32C, mask(c) = rotr(c,15) + 0xa282ead8).  Each payload is a serialized tf.Example with features
 * 'image' (bytes: raw uint8 HWC), 'label', 'height', 'width' (int64).
 *
 * Host only: nothing here touches the GPU; buffers are caller-owned host memory (pinned or not).  Status codes and
 * tg_last_error_string() as in tg_kernels.h.  The value scaling (x/255*2-1, MNIST x/255) and the one-hot encoding of
 * the parser run on the device: tg_u8_affine_f32 / tg_onehot_i32_f32 in tg_kernels.h.
 */
#ifndef TG_IO_H
#define TG_IO_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* CRC-32C (Castagnoli) of n bytes, and TensorFlow's masked form of it. */
uint32_t tg_crc32c(const void* data, int64_t n);
uint32_t tg_crc32c_masked(const void* data, int64_t n);

/* Write (append != 0: append) n images [n][h][w][c] uint8 with their labels as one tf.Example per record. */
int tg_tfrecord_write(const char* path, const uint8_t* images, const int64_t* labels, int64_t n, int h, int w, int c, int append);

/* Frame ONE arbitrary payload as a TFRecord record and write / append it (TensorBoard event files, Training/Summary.py, are
 * TFRecord files of Event protos). */
int tg_record_append(const char* path, const void* payload, int64_t len, int append);

/* Parse one serialized tf.Example: pointers INTO rec for the image bytes; label / height / width values.
 * Missing features are an error (tf.FixedLenFeature without default). */
int tg_example_parse(const uint8_t* rec, int64_t len, const uint8_t** image, int64_t* image_len, int64_t* label, int64_t* height,
                     int64_t* width);

/* Open a TFRecord file: memory-map it, index every record and verify BOTH CRCs of every record (a corrupt or truncated
 * file fails here, with the byte offset in the error string).  An empty file is a dataset of size 0. */
int tg_ds_open(const char* path, void** handle);
int64_t tg_ds_size(void* handle);
/* image geometry of record 0: height, width from its features, channels = image bytes / (height*width). */
int tg_ds_shape(void* handle, int* h, int* w, int* c);
/* raw payload of record i (view into the mapping, valid until tg_ds_close). */
int tg_ds_record(void* handle, int64_t i, const uint8_t** payload, int64_t* len);
/* Decode records idx[0..n) into images [n][h*w*c] uint8 and labels [n] int32 using n_threads host threads (<= 1: the calling
 * thread).  Every record must have the geometry of tg_ds_shape. */
int tg_ds_gather(void* handle, const int64_t* idx, int64_t n, uint8_t* images, int32_t* labels, int n_threads);
int tg_ds_close(void* handle);

#ifdef __cplusplus
}
#endif
#endif
