// How fast can a process read n small files from the page cache?  (the read stage of melf_jpeg_process_files)
//   g++ -O2 -pthread -o /tmp/file_reads tools/ubench/file_reads.cpp
//   ls dir/*.jpg | /tmp/file_reads <threads> <mode>     mode 0: stat pass + open/read/close pass (what the library does)
//                                                        mode 1: one pass, open + fstat + bump-allocated arena + read
//                                                        mode 2: one pass, open + mmap(MAP_POPULATE) + touch + munmap
//                                                        mode 3: as mode 1 with openat(fd of the file's directory, base name): one path
//                                                                component per open instead of the whole path (round 5)
//                                                        mode 4: mode 3 without the fstat (read until EOF into a slot of 64 KiB)
//                                                        mode 5: open + close only; mode 6: openat + close only
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <sys/mman.h>
int main(int argc, char** argv) {
    int nth = atoi(argv[1]); int mode = atoi(argv[2]);
    std::vector<std::string> paths; char buf[4096];
    while (fgets(buf, sizeof buf, stdin)) { std::string s(buf); while (!s.empty() && s.back()=='\n') s.pop_back(); paths.push_back(s); }
    int n = paths.size();
    std::vector<size_t> off(n+1, 0);
    std::vector<uint8_t> arena(256u<<20);
    for (int rep = 0; rep < 6; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        std::atomic<int> next{0};
        auto run = [&](auto f) { next = 0; std::vector<std::thread> th; for (int t = 0; t < nth; ++t) th.emplace_back([&]{ for (;;) { int i = next.fetch_add(8); if (i >= n) break; for (int k = i; k < i+8 && k < n; ++k) f(k);} }); for (auto& t : th) t.join(); };
        std::atomic<size_t> bump{0};
        if (mode == 0) {
            run([&](int i){ struct stat sb; stat(paths[i].c_str(), &sb); off[i+1] = sb.st_size; });
            for (int i = 0; i < n; ++i) off[i+1] += off[i];
        }
        auto t1 = std::chrono::steady_clock::now();
        static int dirfd = -1; static std::vector<std::string> base;
        if (mode >= 3 && dirfd < 0) { std::string d = paths[0].substr(0, paths[0].rfind('/')); dirfd = open(d.c_str(), O_RDONLY|O_DIRECTORY|O_CLOEXEC); for (auto& p : paths) base.push_back(p.substr(p.rfind('/') + 1)); }
        if (mode == 5) { run([&](int i){ int fd = open(paths[i].c_str(), O_RDONLY|O_CLOEXEC); close(fd); }); }
        else if (mode == 6) { run([&](int i){ int fd = openat(dirfd, base[i].c_str(), O_RDONLY|O_CLOEXEC); close(fd); }); }
        else if (mode == 4) { run([&](int i){ int fd = openat(dirfd, base[i].c_str(), O_RDONLY|O_CLOEXEC); size_t o = bump.fetch_add(65536), got = 0; for (;;) { ssize_t r = read(fd, arena.data()+o+got, 65536-got); if (r <= 0) break; got += r; } close(fd); }); }
        else
        run([&](int i){ int fd = mode == 3 ? openat(dirfd, base[i].c_str(), O_RDONLY|O_CLOEXEC) : open(paths[i].c_str(), O_RDONLY|O_CLOEXEC); size_t sz, o;
            if (mode == 2) { struct stat sb; fstat(fd, &sb); void* p = mmap(nullptr, sb.st_size, PROT_READ, MAP_PRIVATE|MAP_POPULATE, fd, 0); close(fd); volatile uint8_t x = 0; for (size_t q = 0; q < (size_t)sb.st_size; q += 4096) x += ((uint8_t*)p)[q]; munmap(p, sb.st_size); return; }
            if (mode == 0) { sz = off[i+1]-off[i]; o = off[i]; } else { struct stat sb; fstat(fd, &sb); sz = sb.st_size; o = bump.fetch_add((sz+63)&~63); }
            size_t got = 0; while (got < sz) { ssize_t r = read(fd, arena.data()+o+got, sz-got); if (r <= 0) break; got += r; } close(fd); });
        auto t2 = std::chrono::steady_clock::now();
        printf("n=%d threads=%d mode=%d: sizes %.2f ms, read %.2f ms\n", n, nth, mode, std::chrono::duration<double,std::milli>(t1-t0).count(), std::chrono::duration<double,std::milli>(t2-t1).count());
    }
}
