// Host side of create_dataset's writer threads (no kernel in this file): widen float32 samples to float64 and write each as one
// `.pt` file = container prefix + payload + per-sample suffix (fast_pt.PtTemplate: the bytes th.save would write -- the reference's
// `th.save(magn_phase.to(th.float64), ...)`, /root/reference/music_gan/create_dataset.py:52-62), plus the float32 side-car row.
// Called through ctypes, which releases the interpreter lock for the duration: with the per-sample work in Python 16 writer
// threads took turns on the lock (4.0 ms of thread time per sample against 0.85 ms for the same system calls made from C).
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/uio.h>
#include <unistd.h>

#include <chrono>
#include <string>

#include "mg_common.h"

namespace {

bool write_all(int fd, struct iovec* iov, int cnt) {
  while (cnt > 0) {
    ssize_t w = writev(fd, iov, cnt);
    if (w < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    if (w == 0) {  // (no progress: a full device reports ENOSPC on the next call at the latest; never spin)
      errno = ENOSPC;
      return false;
    }
    while (cnt > 0 && (size_t)w >= iov->iov_len) {
      w -= iov->iov_len;
      ++iov;
      --cnt;
    }
    if (cnt > 0) {
      iov->iov_base = static_cast<char*>(iov->iov_base) + w;
      iov->iov_len -= w;
    }
  }
  return true;
}

bool pwrite_all(int fd, const char* p, size_t n, off_t off) {
  while (n > 0) {
    ssize_t w = pwrite(fd, p, n, off);
    if (w < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    if (w == 0) {
      errno = ENOSPC;
      return false;
    }
    p += w;
    n -= w;
    off += w;
  }
  return true;
}

}  // namespace

extern "C" int mg_pt_write_samples(const float* rows, int n, int64_t row_floats, const char* paths, const unsigned char* prefix,
                                   int64_t prefix_len, const unsigned char* suffixes, int64_t suffix_len, int side_fd,
                                   int64_t side_off) {
  MG_CHECK_ARG(rows && n > 0 && row_floats > 0 && paths && prefix && suffixes && prefix_len >= 0 && suffix_len >= 0,
               "mg_pt_write_samples: bad arguments");
  double* wide = static_cast<double*>(malloc((size_t)row_floats * sizeof(double)));
  MG_CHECK_ARG(wide != nullptr, "mg_pt_write_samples: out of memory");
  const char* path = paths;
  int rc = MG_OK;
  for (int i = 0; i < n && rc == MG_OK; ++i) {
    const float* src = rows + (size_t)i * row_floats;
    if (side_fd >= 0 && !pwrite_all(side_fd, reinterpret_cast<const char*>(src), (size_t)row_floats * 4, (off_t)(side_off + (int64_t)i * row_floats * 4))) {
      mg_set_error("mg_pt_write_samples: side-car write failed: %s", strerror(errno));
      rc = MG_EIO;
      break;
    }
    for (int64_t k = 0; k < row_floats; ++k) wide[k] = (double)src[k];
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) {
      mg_set_error("mg_pt_write_samples: cannot open %s: %s", path, strerror(errno));
      rc = MG_EIO;
      break;
    }
    struct iovec iov[3] = {{const_cast<unsigned char*>(prefix), (size_t)prefix_len},
                           {wide, (size_t)row_floats * sizeof(double)},
                           {const_cast<unsigned char*>(suffixes + (size_t)i * suffix_len), (size_t)suffix_len}};
    if (!write_all(fd, iov, 3)) {
      mg_set_error("mg_pt_write_samples: write to %s failed: %s", path, strerror(errno));
      rc = MG_EIO;
    }
    if (close(fd) != 0 && rc == MG_OK) {  // (deferred write errors -- quota, a network file system -- surface here)
      mg_set_error("mg_pt_write_samples: closing %s failed: %s", path, strerror(errno));
      rc = MG_EIO;
    }
    if (rc != MG_OK) unlink(path);  // never leave a truncated magn_phase_*.pt behind for AudioDataset to load
    path += strlen(path) + 1;
  }
  free(wide);
  return rc;
}

// Measurement helper of bench.py's host-io bound (create_dataset end to end): what ONE writer thread's sample costs the host whatever
// the product code around it -- n times { a new file <dir>/probe_raw_<tid>_<i>.bin: open, write(file_bytes from a zero buffer), close;
// pwrite(side_bytes) into <dir>/probe_rawside_<tid>.bin } and n times { row_floats float32 -> float64 into a scratch row } -- as plain
// system calls / one loop, timed separately.  Called from several Python threads at once through ctypes (no interpreter lock): a
// Python-level probe of the same calls measured the lock hand-over between threads, not the file system (2x slower than the product).
// The files are left for the caller to remove.
extern "C" int mg_host_io_probe(const char* dir, int tid, int n, int64_t file_bytes, int64_t side_bytes, const float* src,
                                int64_t row_floats, double* write_seconds, double* widen_seconds) {
  MG_CHECK_ARG(dir && n > 0 && file_bytes > 0 && side_bytes >= 0 && src && row_floats > 0 && write_seconds && widen_seconds,
               "mg_host_io_probe: bad arguments");
  const size_t zb = (size_t)(file_bytes > side_bytes ? file_bytes : side_bytes);
  char* zeros = static_cast<char*>(calloc(zb, 1));
  double* wide = static_cast<double*>(malloc((size_t)row_floats * sizeof(double)));
  MG_CHECK_ARG(zeros && wide, "mg_host_io_probe: out of memory");
  int rc = MG_OK;
  const std::string base(dir);
  const std::string side_path = base + "/probe_rawside_" + std::to_string(tid) + ".bin";
  const int side = open(side_path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (side < 0) {
    mg_set_error("mg_host_io_probe: cannot open %s: %s", side_path.c_str(), strerror(errno));
    rc = MG_EIO;
  }
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n && rc == MG_OK; ++i) {
    const std::string path = base + "/probe_raw_" + std::to_string(tid) + "_" + std::to_string(i) + ".bin";
    const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0 || !pwrite_all(fd, zeros, (size_t)file_bytes, 0) || close(fd) != 0 ||
        !pwrite_all(side, zeros, (size_t)side_bytes, (off_t)((int64_t)i * side_bytes))) {
      mg_set_error("mg_host_io_probe: write to %s failed: %s", path.c_str(), strerror(errno));
      rc = MG_EIO;
    }
  }
  *write_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (side >= 0) close(side);
  t0 = std::chrono::steady_clock::now();
  double sink = 0.0;
  for (int i = 0; i < n; ++i) {
    for (int64_t k = 0; k < row_floats; ++k) wide[k] = (double)src[k];
    sink += wide[(size_t)i % (size_t)row_floats];
  }
  *widen_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() + (sink == 12345.678 ? 1e-12 : 0.0);
  free(zeros);
  free(wide);
  return rc;
}
