// mpf_matgen -- generator of the reference's benchmark input files.
//
// Behavioural restatement of the reference tool (matrix_generator.cpp:6-90): same command line, same file format,
// byte-identical output on glibc hosts, AND on any other libc because the C library's rand() is not used: glibc's
// default generator (TYPE_3 additive feedback, r[i] = r[i-3] + r[i-31], seed 1 since the reference never calls
// srand()) is re-implemented here.
//
//   mpf_matgen filename maxSize [step=2] [exp|lin] [sparsity=0.0]
//
// File layout: a 16-blank header line that is overwritten with the matrix count at the end (matrix_generator.cpp:53,
// :84-85); then for size = 2, 2*step (exp) or 2+step (lin), ... <= maxSize: the size on its own line, `size` rows of
// `size` values "v " each, a blank line.  Values are (rand() % 100) / 10.0; with sparsity > 0 one extra draw per
// element decides whether the value is 0 instead (:63-67).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace {
struct GlibcRand { // glibc random_r TYPE_3, degree 31, separation 3
    int32_t r[31];
    int f = 3, b = 0;
    explicit GlibcRand(unsigned seed) {
        if (seed == 0) seed = 1;
        r[0] = (int32_t)seed;
        for (int i = 1; i < 31; ++i) {
            const int64_t hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
            int64_t w = 16807 * lo - 2836 * hi;
            if (w < 0) w += 2147483647;
            r[i] = (int32_t)w;
        }
        for (int i = 0; i < 310; ++i) next();
    }
    int next() {
        const uint32_t v = (uint32_t)r[f] + (uint32_t)r[b];
        r[f] = (int32_t)v;
        f = (f + 1) % 31;
        b = (b + 1) % 31;
        return (int)(v >> 1);
    }
};

// what `ofstream << double` prints with default formatting (6 significant digits, %g-like)
void put_value(FILE *fp, double v) { std::fprintf(fp, "%g ", v); }
} // namespace

int main(int argc, char **argv) {
    if (argc < 3) {
        std::printf("Usage: %s filename maxSize [step=2] [function=exp (exp/lin)] [sparsity=0.0]\n", argv[0]);
        std::printf("  sparsity: fraction of zeros in the matrix (0.0 = dense, 0.9 = 90%% zeros)\n");
        return -1;
    }
    FILE *fp = std::fopen(argv[1], "wb");
    if (!fp) { std::printf("Failed to open %s\n", argv[1]); return -1; }
    const int max_size = std::atoi(argv[2]);
    if (max_size <= 0) { std::printf("Invalid maxSize: %d\n", max_size); return -1; }
    int step = 2;
    if (argc > 3 && (step = std::atoi(argv[3])) <= 0) { std::printf("Invalid step: %d\n", step); return -1; }
    bool geometric = true;
    if (argc > 4) {
        const std::string fn = argv[4];
        if (fn == "lin") geometric = false;
        else if (fn != "exp") { std::printf("Invalid function: %s. Use 'exp' or 'lin'.\n", fn.c_str()); return -1; }
    }
    double sparsity = 0.0;
    if (argc > 5) {
        sparsity = std::atof(argv[5]);
        if (sparsity < 0.0 || sparsity >= 1.0) { std::printf("Invalid sparsity: %g. Must be in [0.0, 1.0).\n", sparsity); return -1; }
    }
    GlibcRand rng(1);
    std::fputs("                \n", fp);
    int count = 0;
    for (int size = 2; size <= max_size; size = geometric ? size * step : size + step) {
        std::fprintf(fp, "%d\n", size);
        for (int i = 0; i < size; ++i) {
            for (int j = 0; j < size; ++j) {
                double v;
                if (sparsity > 0.0 && (double)rng.next() / (2147483647.0 + 1.0) < sparsity) v = 0.0;
                else v = (double)(rng.next() % 100) / 10.0;
                put_value(fp, v);
            }
            std::fputc('\n', fp);
        }
        std::fputc('\n', fp);
        ++count;
        std::printf("Generating matrix of size %d\r", geometric ? size * step : size + step);
        std::fflush(stdout);
        if (geometric && step == 1) break; // size *= 1 would never end (the reference loops forever here)
    }
    std::fseek(fp, 0, SEEK_SET);
    std::fprintf(fp, "%d", count);
    std::fclose(fp);
    std::printf("\nnumber of matrices: %d\n", count);
    return 0;
}
