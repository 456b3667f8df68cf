/**
 * @file mapped_file.hxx
 * @brief Read-only view of a whole file: mmap + sequential-access advice, falling back to a
 * plain read into a buffer when mapping is refused (pipes, odd filesystems).
 * (Role of the reference's container/detail/mapped_file.hxx:78-124.)
 */
#pragma once

#include <cstddef>
#include <cstdio>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <loops/error.hxx>

namespace loops {
namespace detail {

class mapped_file_t {
 public:
  explicit mapped_file_t(const std::string& path) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    error::throw_if_exception(fd < 0, "mapped_file_t: cannot open " + path);
    struct stat st {};
    if (::fstat(fd, &st) != 0) {
      ::close(fd);
      error::throw_if_exception(true, "mapped_file_t: cannot stat " + path);
    }
    size_ = static_cast<std::size_t>(st.st_size);
    if (size_ > 0) {
      void* p = ::mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd, 0);
      if (p != MAP_FAILED) {
        ::madvise(p, size_, MADV_WILLNEED);  // (read by several threads at once: whole-file read-ahead, not one sequential stream)
        base_ = static_cast<const char*>(p);
        mapped_ = true;
      } else {  // fall back to read()
        buffer_.resize(size_);
        std::size_t got = 0;
        while (got < size_) {
          const ssize_t n = ::read(fd, buffer_.data() + got, size_ - got);
          if (n <= 0) break;
          got += static_cast<std::size_t>(n);
        }
        size_ = got;
        base_ = buffer_.data();
      }
    }
    ::close(fd);
  }
  mapped_file_t(const mapped_file_t&) = delete;
  mapped_file_t& operator=(const mapped_file_t&) = delete;
  ~mapped_file_t() {
    if (mapped_) ::munmap(const_cast<char*>(base_), size_);
  }

  const char* data() const { return base_; }
  const char* end() const { return base_ + size_; }
  std::size_t size() const { return size_; }

 private:
  const char* base_ = nullptr;
  std::size_t size_ = 0;
  bool mapped_ = false;
  std::vector<char> buffer_;
};

}  // namespace detail
}  // namespace loops
