// host-only harness: the symbolic analysis (ordering + tree + supernodes + front rows) of a 2-D grid under ThreadSanitizer
#include "symbolic.hpp"
#include <cstdio>
#include <vector>
using namespace kvx;
int main(int argc, char **argv)
{
    const int64_t g = argc > 1 ? atoll(argv[1]) : 300;
    const int64_t n = g * g;
    std::vector<int64_t> cp(n + 1, 0), ri;
    std::vector<double> vx;
    for (int64_t j = 0; j < n; j++) {
        const int64_t x = j % g, y = j / g;
        ri.push_back(j);
        if (x + 1 < g) ri.push_back(j + 1);
        if (y + 1 < g) ri.push_back(j + g);
        cp[j + 1] = (int64_t)ri.size();
    }
    for (int rep = 0; rep < 2; rep++) {
        Symbolic S;
        SymOpts o;
        o.ordering = rep == 0 ? 2 : 0;
        analyze(n, cp.data(), ri.data(), 'L', nullptr, o, S);
        printf("rep %d: nsuper %ld lnz %ld levels %d\n", rep, (long)S.nsuper, (long)S.lnz, (int)S.nlevels);
    }
    return 0;
}
