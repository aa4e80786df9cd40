// decompressor.hpp -- C++ host-side mirror of the reference's only public item,
//
//     pub struct Decompressor<R: Read>                         (reference src/lib.rs:377-394)
//     pub fn new(r: R) -> Decompressor<R>                      (reference src/lib.rs:398-410)
//     impl<R: Read> Read for Decompressor<R> { fn read(..) }   (reference src/lib.rs:2173-2193)
//
// on top of the C ABI (include/brx.h).  The reference is Rust; this image has no Rust toolchain, so the
// host side that sits above the C ABI is written in C++ with the same shape (INTEGRATION.md shows the Rust
// shim a maintainer of the reference would add).  Header-only; link with -lbrx.
//
//   brotli::Decompressor<brotli::SliceReader> d(brotli::SliceReader(ptr, n));
//   std::vector<uint8_t> out = d.read_to_end();        // == Read::read_to_end
//
// Semantics of read(): n > 0 bytes, 0 at end of stream forever after; an invalid stream throws
// brotli::InvalidData whose what() is the reference's description string (the reference returns
// io::Error::new(ErrorKind::InvalidData, description), src/lib.rs:2177).  Two documented differences of the
// GPU backend: the inner reader of a SHORT stream (compressed input < 4 MiB) is drained eagerly on the first read, a longer one is
// pulled as it decodes -- in bounded device memory whatever the stream holds: the reader's windows grow when ONE command produces
// more than the room behind the output window or ONE item needs more input than the input window (brx.h, brx_stream_new_reader;
// before round 5 such a stream failed over a reader), and the pull callback runs with the context's lock released, so R may itself
// be a Decompressor on default_context(); for a stream that fails the bytes
// produced before the error are delivered first, then the error (the reference delivers an unspecified prefix too,
// SURVEY.md Q13).
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/brx.h"

namespace brotli {

struct InvalidData : std::runtime_error {
    int status;
    InvalidData(int st, const char *what) : std::runtime_error(what), status(st) {}
};

// A minimal "Read": any type with  size_t read(uint8_t *buf, size_t len)  returning 0 at EOF works as R.
class SliceReader {
    const uint8_t *p_;
    size_t n_, at_ = 0;

  public:
    SliceReader(const uint8_t *p, size_t n) : p_(p), n_(n) {}
    size_t read(uint8_t *buf, size_t len) {
        size_t k = n_ - at_ < len ? n_ - at_ : len;
        for (size_t i = 0; i < k; i++) buf[i] = p_[at_ + i];
        at_ += k;
        return k;
    }
};

inline brx_ctx *default_context() {
    static brx_ctx *ctx = [] {
        brx_ctx *c = nullptr;
        int rc = brx_ctx_create(&c, 0);
        if (rc != BRX_SUCCESS) throw std::runtime_error(std::string("brx_ctx_create: ") + brx_last_error());
        return c;
    }();
    return ctx;
}

// Decompressor<R>: small streams are drained and POOLED (the first read of any queued Decompressor decodes all of them in one
// batch); a stream whose compressed input does not end within the first 4 MiB is decoded the way the reference does it -- the
// inner reader is PULLED while the stream decodes (brx_stream_new_reader: src/bitreader/mod.rs:21-53), compressed input and
// decoded output both in bounded sliding windows on the device, whatever the stream's length.
template <class R> class Decompressor {
    R inner_;
    brx_stream *stream_ = nullptr;
    std::vector<uint8_t> head_; // what was drained before the stream turned out to be a long one: replayed to the puller first
    size_t head_at_ = 0;

    static size_t pull(void *user, uint8_t *buf, size_t cap) { // brx_read_fn
        Decompressor *d = static_cast<Decompressor *>(user);
        try {
            if (d->head_at_ < d->head_.size()) {
                size_t k = d->head_.size() - d->head_at_ < cap ? d->head_.size() - d->head_at_ : cap;
                for (size_t i = 0; i < k; i++) buf[i] = d->head_[d->head_at_ + i];
                d->head_at_ += k;
                if (d->head_at_ == d->head_.size()) std::vector<uint8_t>().swap(d->head_), d->head_at_ = 0;
                return k;
            }
            return d->inner_.read(buf, cap);
        } catch (...) {
            return 0; // no exception crosses the C ABI: the input ends here
        }
    }

  public:
    static constexpr size_t POOLED_LIMIT = 4u << 20;
    explicit Decompressor(R r) : inner_(std::move(r)) {} // infallible and reads nothing, like the reference's new()
    Decompressor(const Decompressor &) = delete;          // the reference derives Debug only, not Clone
    Decompressor &operator=(const Decompressor &) = delete;
    ~Decompressor() { brx_stream_free(stream_); }

    // Drain the inner reader (up to POOLED_LIMIT) and queue the stream on the context without decoding: the first read() of ANY
    // queued Decompressor then decodes all of them in one batch (brx.h, Read facade).  A longer input becomes a pulled stream.
    void prepare() {
        if (stream_) return;
        std::vector<uint8_t> in;
        uint8_t tmp[65536];
        bool ended = false;
        while (in.size() < POOLED_LIMIT) {
            size_t k = inner_.read(tmp, sizeof tmp);
            if (k == 0) { ended = true; break; }
            in.insert(in.end(), tmp, tmp + k);
        }
        if (ended) {
            stream_ = brx_stream_new(default_context(), in.data(), in.size());
        } else {
            head_.swap(in);
            stream_ = brx_stream_new_reader(default_context(), &Decompressor::pull, this);
        }
        if (!stream_) throw std::runtime_error("brx_stream_new failed");
    }

    size_t read(uint8_t *buf, size_t len) {
        prepare();
        int64_t n = brx_stream_read(stream_, buf, len);
        if (n < -900) throw std::runtime_error(std::string("libbrx: ") + brx_last_error());
        if (n < 0) throw InvalidData((int)-n, brx_status_str((int32_t)-n));
        return (size_t)n;
    }

    std::vector<uint8_t> read_to_end() {
        std::vector<uint8_t> out;
        uint8_t tmp[65536];
        for (size_t k; (k = read(tmp, sizeof tmp)) > 0;) out.insert(out.end(), tmp, tmp + k);
        return out;
    }
};

} // namespace brotli
