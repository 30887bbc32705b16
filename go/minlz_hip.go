//go:build hip && cgo

// Package minlz — HIP back end for the block codec (drop this file into github.com/minio/minlz).
//
// It binds the seams INTEGRATION.md section 1 lists to libminlz_hip.so (include/minlz_hip.h):
//
//	WriterCustomEncoder contract  (writer.go:1293-1304, consulted at writer.go:566-570)  -> mlz_encode_block
//	minLZDecode build-tag seam    (decode_amd64.go:28-36 / decode_other.go:24-26)        -> mlz_decode_block
//	Encode / Decode               (encode.go:74, decode.go:50)                            -> mlz_encode / mlz_decode
//	EncodeBuffer + Close / Reader (writer.go:441-563,1051-1126; reader.go:248-543)        -> mlz_stream_encode / mlz_stream_decode
//
// Build:  CGO_ENABLED=1 go build -tags hip   (the default build, CGO_ENABLED=0, never sees this file).
// Under the hip tag decode_amd64.go / decode_other.go need `&& !hip` added to their build constraints, because this
// file defines minLZDecode.  Go is not installed in the image this repository is built in: the file is kept
// `go vet`-clean by inspection only; every C entry point it calls is exercised through the same ABI by tests/.
package minlz

/*
#cgo CFLAGS: -I${SRCDIR}/include
#cgo LDFLAGS: -L${SRCDIR}/lib -lminlz_hip -Wl,-rpath,${SRCDIR}/lib
#include <stdint.h>
#include <stddef.h>
#include "minlz_hip.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"io"
	"sync"
	"sync/atomic"
	"unsafe"
)

// ErrHIP reports a HIP runtime failure (no device, launch error, grid barrier timed out).
// Callers fall back to the CPU path; the shim below does so itself wherever the reference has one.
var ErrHIP = errors.New("minlz: HIP runtime failure")

// ErrDstTooSmall reports that a caller-supplied buffer cannot hold the result (the shim sizes its own buffers,
// so this only surfaces for internal mistakes).
var ErrDstTooSmall = errors.New("minlz: destination buffer too small")

var (
	hipOnce sync.Once
	hipCtx  *C.mlz_ctx
)

// hipContext returns the process-wide context, or nil when no device is usable.
//
// It spans EVERY visible GPU (mlz_init_devices with a nil list; restrict it with HIP_VISIBLE_DEVICES): a Go program is one process,
// so the fan-out of a stream's blocks over the GPUs of a node lives behind the C ABI — mlz_stream_encode / mlz_stream_decode and the
// batch calls deal contiguous block ranges to the devices, each over its own PCIe link, and the results land in the caller's buffer at
// their final offsets (no collective: the consumer is host memory).  Single-block calls (the Writer's goroutine per block,
// writer.go:501-560; the Reader's, reader.go:830-859) go to the devices in turn, each device with its own combining queue.
//
// One context serializes the calls that reach the SAME device (a device owns one workspace): two Writers in one process share the
// devices' queues.  That is the intended behaviour for a node-wide back end — the queue batches their blocks into common launches —;
// a program that wants two independent pipelines on one GPU lists the device twice (mlz_init_devices({0, 0}, 2, ...)).
func hipContext() *C.mlz_ctx {
	hipOnce.Do(func() {
		var c *C.mlz_ctx
		if C.mlz_init_devices(nil, 0, &c) == 0 {
			hipCtx = c
		}
	})
	return hipCtx
}

// HIPDevices reports how many GPUs the back end deals blocks to (0 = no usable device: every call takes the CPU path).
func HIPDevices() int {
	c := hipContext()
	if c == nil {
		return 0
	}
	return int(C.mlz_device_count(c))
}

// hipError maps -MLZ_ERR_* (include/minlz_hip.h) to the package's errors.
func hipError(code int) error {
	switch code {
	case C.MLZ_ERR_CORRUPT:
		return ErrCorrupt
	case C.MLZ_ERR_TOO_LARGE:
		return ErrTooLarge
	case C.MLZ_ERR_UNSUPPORTED:
		return ErrUnsupported
	case C.MLZ_ERR_INVALID_LEVEL:
		return ErrInvalidLevel
	case C.MLZ_ERR_CRC:
		return ErrCRC
	case C.MLZ_ERR_DST_TOO_SMALL:
		return ErrDstTooSmall
	case C.MLZ_ERR_HIP:
		return ErrHIP
	default:
		return fmt.Errorf("minlz: hip back end error %d", code)
	}
}

func bytePtr(b []byte) *C.uint8_t {
	if len(b) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&b[0]))
}

// hipEncodeBlock has the WriterCustomEncoder contract (writer.go:1293-1304): tokens only into dst;
// > 0 bytes written, 0 = incompressible (the Writer emits an uncompressed chunk), < 0 = decline
// (the built-in encoder runs).
func hipEncodeBlock(level int) func(dst, src []byte) int {
	return func(dst, src []byte) int {
		c := hipContext()
		if c == nil || len(src) == 0 || len(dst) < len(src) {
			return -1
		}
		n := C.mlz_encode_block(c, C.int(level), bytePtr(src), C.size_t(len(src)), bytePtr(dst), C.size_t(len(dst)))
		if n < 0 {
			if int(-n) == C.MLZ_ERR_HIP {
				atomic.AddUint64(&hipFallbacks, 1)
			}
			return -1 // HIP failure or a level the device does not serve: let the built-in encoder run
		}
		return int(n)
	}
}

// WriterHIP moves the Writer's block encoder to the GPU for the levels the device serves
// (LevelSuperFast, LevelFastest, LevelBalanced); LevelSmallest keeps the CPU encoder.
// It reads the level the preceding options selected, so pass it after WriterLevel:
//
//	w := minlz.NewWriter(dst, minlz.WriterLevel(minlz.LevelFastest), minlz.WriterBlockSize(8<<20), minlz.WriterHIP())
//
// The Writer still calls the encoder from one goroutine per block (writer.go:501-560); the library combines
// concurrent single-block calls into batched launches.
func WriterHIP() WriterOption {
	return func(w *Writer) error {
		lvl := int(w.level)
		if lvl == LevelUncompressed || lvl > LevelBalanced || lvl < LevelSuperFast {
			return nil
		}
		if hipContext() == nil {
			return nil // no device: the Writer behaves exactly as without this option
		}
		w.customEnc = hipEncodeBlock(lvl)
		return nil
	}
}

// NewHIPWriter is NewWriter with WriterHIP appended.
func NewHIPWriter(w io.Writer, opts ...WriterOption) *Writer {
	all := make([]WriterOption, 0, len(opts)+1)
	all = append(all, opts...)
	all = append(all, WriterHIP())
	return NewWriter(w, all...)
}

// minLZDecode replaces decode_amd64.go / decode_other.go under the hip tag (callers: decode.go:74,
// reader.go:334,833,1120).  0 = ok, decodeErrCodeCorrupt otherwise (decode.go:26).
func minLZDecode(dst, src []byte) int {
	c := hipContext()
	if c == nil || len(dst) == 0 {
		return minLZDecodeGo(dst, src)
	}
	r := C.mlz_decode_block(c, bytePtr(src), C.size_t(len(src)), bytePtr(dst), C.size_t(len(dst)))
	if r < 0 { // runtime failure (never a verdict on the input: those are 0 / 1): CPU path, and counted
		atomic.AddUint64(&hipFallbacks, 1)
		return minLZDecodeGo(dst, src)
	}
	return int(r)
}

// hipFallbacks counts the block calls that reached the device and came back with a device failure, so that the caller
// decoded (or encoded) on the CPU instead.  A healthy device leaves it at 0: a monitoring hook for a failing GPU that the
// silent fallback would otherwise hide.
var hipFallbacks uint64

// HIPFallbacks returns the number of device failures that were served by the CPU path since the process started.
func HIPFallbacks() uint64 { return atomic.LoadUint64(&hipFallbacks) }

// HIPEncode is Encode (encode.go:74-139) with the whole block — header included — built on the device.
func HIPEncode(dst, src []byte, level int) ([]byte, error) {
	c := hipContext()
	if c == nil {
		return Encode(dst, src, level)
	}
	n := MaxEncodedLen(len(src))
	if n < 0 {
		return nil, ErrTooLarge
	}
	if cap(dst) < n {
		dst = make([]byte, n)
	}
	dst = dst[:n]
	r := C.mlz_encode(c, C.int(level), bytePtr(src), C.size_t(len(src)), bytePtr(dst), C.size_t(len(dst)))
	if r < 0 {
		if int(-r) == C.MLZ_ERR_HIP || int(-r) == C.MLZ_ERR_INVALID_LEVEL {
			if int(-r) == C.MLZ_ERR_HIP {
				atomic.AddUint64(&hipFallbacks, 1)
			}
			return Encode(dst, src, level) // device failure, or LevelSmallest: CPU encoder
		}
		return nil, hipError(int(-r))
	}
	return dst[:r], nil
}

// HIPDecode is Decode (decode.go:50-115) for MinLZ blocks; Snappy/S2 blocks and device failures go to Decode.
func HIPDecode(dst, src []byte) ([]byte, error) {
	c := hipContext()
	if c == nil || len(src) == 0 || src[0] != 0 {
		return Decode(dst, src)
	}
	dl := C.mlz_decoded_len(bytePtr(src), C.size_t(len(src)))
	if dl < 0 {
		return nil, hipError(int(-dl))
	}
	if cap(dst) < int(dl) {
		dst = make([]byte, int(dl))
	}
	dst = dst[:int(dl)]
	r := C.mlz_decode(c, bytePtr(src), C.size_t(len(src)), bytePtr(dst), C.size_t(len(dst)))
	if r < 0 {
		if int(-r) == C.MLZ_ERR_HIP {
			return Decode(dst, src)
		}
		return nil, hipError(int(-r))
	}
	return dst[:r], nil
}

// HIPEncodeStream is NewWriter(dst, WriterLevel(level), WriterBlockSize(blockSize), WriterAddIndex(addIndex))
// + EncodeBuffer(src) + Close() in one call: stream header, chunks, CRCs, EOF and index are made by the library.
func HIPEncodeStream(dst, src []byte, level, blockSize int, addIndex bool) ([]byte, error) {
	c := hipContext()
	if c == nil {
		return nil, ErrHIP
	}
	var flags C.uint32_t
	if addIndex {
		flags |= C.MLZ_STREAM_ADD_INDEX
	}
	need := int(C.mlz_stream_bound(C.uint64_t(len(src)), C.uint32_t(blockSize), flags))
	if cap(dst) < need {
		dst = make([]byte, need)
	}
	dst = dst[:need]
	n := C.mlz_stream_encode(c, C.int(level), C.uint32_t(blockSize), flags, bytePtr(src), C.size_t(len(src)), bytePtr(dst), C.size_t(len(dst)))
	if n < 0 {
		return nil, hipError(int(-n))
	}
	return dst[:n], nil
}

// HIPDecodeStream reads a complete MinLZ stream to EOF (Reader semantics, reader.go:248-543).
func HIPDecodeStream(dst, src []byte, ignoreCRC bool) ([]byte, error) {
	c := hipContext()
	if c == nil {
		return nil, ErrHIP
	}
	dl := C.mlz_stream_decoded_len(bytePtr(src), C.size_t(len(src)))
	if dl < 0 {
		return nil, hipError(int(-dl))
	}
	if cap(dst) < int(dl) {
		dst = make([]byte, int(dl))
	}
	dst = dst[:int(dl)]
	var flags C.uint32_t
	if ignoreCRC {
		flags |= C.MLZ_STREAM_IGNORE_CRC
	}
	n := C.mlz_stream_decode(c, flags, bytePtr(src), C.size_t(len(src)), bytePtr(dst), C.size_t(len(dst)))
	if n < 0 {
		return nil, hipError(int(-n))
	}
	return dst[:n], nil
}
