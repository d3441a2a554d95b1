// oxDNA text trajectories: native reader behind mythos_oxdna_read_trajectory (host code only).
//
// Replaces the line-by-line Python parse of the reference (mythos/input/trajectory.py:192-320: every frame is
//   t = <time> / b = <bx> <by> <bz> / E = <e1> <e2> <e3>
// followed by one line of 15 numbers per nucleotide: com(3) a1(3) a3(3) v(3) L(3)).  The file is read once, the
// frame headers are located by a line scan, and the frames are parsed concurrently with std::from_chars.
// DiffTRe reweighting of externally generated trajectories is otherwise dominated by this
// parse for large systems.
#include <algorithm>
#include <atomic>
#include <charconv>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "mythos_internal.h"

namespace {

// read-only mapping of the whole file (no copy, no zero fill)
struct MappedFile {
  const char* data = nullptr;
  size_t size = 0;
  bool ok = false;
  explicit MappedFile(const char* path) {
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return;
    struct stat st;
    if (::fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
      ::close(fd);
      return;
    }
    size = (size_t)st.st_size;
    ok = true;
    if (size) {
      void* m = ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
      if (m == MAP_FAILED) {
        ok = false;
        size = 0;
      } else {
        data = (const char*)m;
        ::madvise(m, size, MADV_SEQUENTIAL);
      }
    }
    ::close(fd);
  }
  ~MappedFile() {
    if (data) ::munmap((void*)data, size);
  }
  MappedFile(const MappedFile&) = delete;
  MappedFile& operator=(const MappedFile&) = delete;
};

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }

// next number in [p, end); advances p past it
inline bool next_number(const char*& p, const char* end, double& v) {
  while (p < end && is_space(*p)) ++p;
  if (p < end && *p == '+') ++p;
  const auto r = std::from_chars(p, end, v);
  if (r.ec != std::errc() || r.ptr == p) return false;
  p = r.ptr;
  return true;
}

// "<key> = v0 .. v(count-1)" on the line at p; advances p to the next line
bool header_line(const char*& p, const char* end, char key, int count, double* out) {
  while (p < end && is_space(*p)) ++p;
  if (p >= end || *p != key) return false;
  const char* nl = (const char*)std::memchr(p, '\n', (size_t)(end - p));
  const char* stop = nl ? nl : end;
  const char* eq = (const char*)std::memchr(p, '=', (size_t)(stop - p));
  if (!eq) return false;
  const char* q = eq + 1;
  for (int k = 0; k < count; ++k)
    if (!next_number(q, stop, out[k])) return false;
  p = nl ? nl + 1 : end;
  return true;
}

struct FrameError {
  int frame = -1;
  int kind = 0;  // 1 header, 2 too few rows, 3 too many rows
};

// parses the frame whose header starts at [p, end) (end = next header or end of file)
int parse_frame(const char* p, const char* end, int n, double* t, double* b, double* e, double* rows) {
  if (!header_line(p, end, 't', 1, t) || !header_line(p, end, 'b', 3, b) || !header_line(p, end, 'E', 3, e)) return 1;
  const size_t count = (size_t)n * 15;
  for (size_t k = 0; k < count; ++k)
    if (!next_number(p, end, rows[k])) return 2;
  while (p < end && is_space(*p)) ++p;
  return p < end ? 3 : 0;
}

}  // namespace

extern "C" {

int mythos_oxdna_read_trajectory(const char* path, int n, int max_frames, double* times, double* box, double* energies,
                                 double* frames, int* n_frames) {
  if (!path || n < 1 || !n_frames || max_frames < 0) {
    mythos::set_error("mythos_oxdna_read_trajectory: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  const MappedFile file(path);
  if (!file.ok) {
    mythos::set_error(std::string("mythos_oxdna_read_trajectory: cannot read ") + path);
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  const char* base = file.data;
  const char* end = base + file.size;
  // frame headers: lines whose first character is 't'
  std::vector<const char*> starts;
  {
    const char* p = base;
    bool first_content = true;
    while (p < end) {
      if (*p == 't') starts.push_back(p);
      else if (first_content && !is_space(*p)) {
        mythos::set_error("mythos_oxdna_read_trajectory: the file does not start with a 't = ...' line");
        return MYTHOS_ERR_INVALID_ARGUMENT;
      }
      if (!is_space(*p)) first_content = false;
      const char* nl = (const char*)std::memchr(p, '\n', (size_t)(end - p));
      if (!nl) break;
      p = nl + 1;
    }
  }
  const int total = (int)starts.size();
  *n_frames = total;
  if (!frames || max_frames == 0) return MYTHOS_OK;
  if (!times || !box || !energies) {
    mythos::set_error("mythos_oxdna_read_trajectory: times, box and energies are required with frames");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  const int todo = std::min(total, max_frames);
  starts.push_back(end);
  std::atomic<int> next{0};
  std::atomic<int> bad_frame{-1}, bad_kind{0};
  auto work = [&]() {
    for (int f = next.fetch_add(1); f < todo; f = next.fetch_add(1)) {
      const int kind = parse_frame(starts[f], starts[f + 1], n, times + f, box + 3 * (size_t)f, energies + 3 * (size_t)f,
                                   frames + (size_t)f * n * 15);
      if (kind) {
        int expect = -1;
        if (bad_frame.compare_exchange_strong(expect, f)) bad_kind.store(kind);
        return;
      }
    }
  };
  const size_t bytes = file.size;
  int threads = (int)std::min<size_t>({(size_t)std::max(1u, std::thread::hardware_concurrency()), (size_t)8, (size_t)todo,
                                       bytes / (4u << 20) + 1});
  if (threads <= 1) {
    work();
  } else {
    std::vector<std::thread> pool;
    for (int k = 0; k < threads; ++k) pool.emplace_back(work);
    for (auto& th : pool) th.join();
  }
  if (bad_frame.load() >= 0) {
    const char* what = bad_kind.load() == 1   ? "has a malformed t/b/E header"
                       : bad_kind.load() == 2 ? "holds fewer nucleotide rows than the strand lengths give"
                                              : "holds more nucleotide rows than the strand lengths give";
    mythos::set_error("mythos_oxdna_read_trajectory: frame " + std::to_string(bad_frame.load()) + " " + what + " (n = " +
                      std::to_string(n) + ")");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  return MYTHOS_OK;
}

// Writer: the text of every frame is produced concurrently into its own buffer (std::to_chars without a precision:
// the shortest text that reads back to the same double, what the reference's str(float) prints -
// mythos/input/trajectory.py:323-331 - so written trajectories restart bit for bit), then the buffers are written in order.  Replaces the reference's per-frame numpy.savetxt
// (mythos/input/trajectory.py:322-331, mythos/simulators/io.py:146-170).
int mythos_oxdna_write_trajectory(const char* path, int n, int n_frames, const double* times, const double* box,
                                  const double* energies, const double* frames, int append) {
  if (!path || n < 1 || n_frames < 0 || (n_frames > 0 && (!times || !box || !energies || !frames))) {
    mythos::set_error("mythos_oxdna_write_trajectory: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  FILE* f = std::fopen(path, append ? "ab" : "wb");
  if (!f) {
    mythos::set_error(std::string("mythos_oxdna_write_trajectory: cannot open ") + path);
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  auto put = [](std::string& out, double v) {
    char tmp[40];
    const auto r = std::to_chars(tmp, tmp + sizeof(tmp), v);  // shortest text that parses back to v exactly
    out.append(tmp, r.ptr);
  };
  auto format_frame = [&](int k, std::string& out) {
    out.clear();
    out.reserve((size_t)n * 15 * 20 + 128);
    out += "t = ";
    put(out, times[k]);
    out += "\nb =";
    for (int a = 0; a < 3; ++a) out += ' ', put(out, box[3 * (size_t)k + a]);
    out += "\nE =";
    for (int a = 0; a < 3; ++a) out += ' ', put(out, energies[3 * (size_t)k + a]);
    out += '\n';
    const double* fr = frames + (size_t)k * n * 15;
    for (int i = 0; i < n; ++i) {
      for (int c = 0; c < 15; ++c) {
        if (c) out += ' ';
        put(out, fr[(size_t)i * 15 + c]);
      }
      out += '\n';
    }
  };
  const size_t frame_bytes = (size_t)n * 15 * 20;
  const int threads = (int)std::min<size_t>({(size_t)std::max(1u, std::thread::hardware_concurrency()), (size_t)8,
                                             (size_t)std::max(n_frames, 1), frame_bytes * (size_t)n_frames / (4u << 20) + 1});
  // batches of `threads` frames: formatted side by side, written in order
  std::vector<std::string> buf((size_t)threads);
  bool ok = true;
  for (int k0 = 0; k0 < n_frames && ok; k0 += threads) {
    const int m = std::min(threads, n_frames - k0);
    if (m == 1) {
      format_frame(k0, buf[0]);
    } else {
      std::vector<std::thread> pool;
      for (int q = 0; q < m; ++q) pool.emplace_back([&, q] { format_frame(k0 + q, buf[(size_t)q]); });
      for (auto& th : pool) th.join();
    }
    for (int q = 0; q < m && ok; ++q) ok = std::fwrite(buf[(size_t)q].data(), 1, buf[(size_t)q].size(), f) == buf[(size_t)q].size();
  }
  if (std::fclose(f) != 0) ok = false;
  if (!ok) {
    mythos::set_error(std::string("mythos_oxdna_write_trajectory: write to ") + path + " failed");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  return MYTHOS_OK;
}

}  // extern "C"
