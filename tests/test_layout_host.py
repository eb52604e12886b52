"""CPU checks of the index maps in nerf_amd/csrc/nerf_layout.h (they are `__host__ __device__ inline`: the pack kernel,
the render kernels, the reduce kernel and the host all call the same functions; on the GPU they are exercised only
through parity).  A small host program includes the header with the qualifiers defined away and walks the invariants:

* colour / class slot map (row_of_slot, slot_of_row, row_of_tile0, class_mask_tile0): for every color_outputs 1..12 and
  row count up to 64, slot -> row is a bijection of the used slots onto 0 .. n_out - 1 that keeps density at slot 0;
* layer-0 feature permutation (layer0_source_feature / layer0_kernel_column) for 1..16 encoding scales and every
  scales-per-lane-group: the two functions are inverses, every feature has exactly one column;
* the dense layer 0 of the 4-tile kernels (layer0_dense_source_slot): dense slots 0..11 are exactly the old slots a lane
  group fills when it evaluates at most two scales — 0..5 (sine) and 12..17 (shifted) — in order."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROGRAM = r"""
#include <cstdio>
#include <cstdlib>
#include <set>
#define __host__
#define __device__
#include "nerf_layout.h"
using namespace nerf_layout;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED line %d: %s\n", __LINE__, #c); std::exit(1); } } while (0)
int main() {
    // ---- output tile: slots <-> rows of the last Linear ----
    for (int colors = 1; colors <= kMaxColors; ++colors)
        for (int n_out = 1 + colors; n_out <= kOutPad; ++n_out) {
            const Shape s{256, 96, n_out, colors};
            CHECK(shape_ok(s));
            std::set<int> rows;
            CHECK(row_of_slot(0, s) == 0);                                  // density
            for (int c = 0; c < colors; ++c) {
                CHECK(is_color_slot(color_slot(c), colors));
                CHECK(row_of_slot(color_slot(c), s) == 1 + c);
            }
            int mask = 0;
            for (int n = 0; n < kOutPad; ++n) {
                const int row = row_of_slot(n, s);
                CHECK(row >= -1 && row < n_out);
                if (row >= 0) {
                    CHECK(rows.insert(row).second);                         // no row twice
                    CHECK(slot_of_row(row, s) == n);
                    if (n < 16) CHECK(row_of_tile0(n / 4, n % 4, colors, n_out) == row);
                    if (n >= 1 && n < 16 && row > colors) mask |= 1 << n;
                }
            }
            CHECK((int)rows.size() == n_out);                               // every row has a slot
            CHECK(class_mask_tile0(s) == mask);
        }
    // ---- layer 0: kernel columns <-> the reference's features ----
    for (int scales = 1; scales <= 16; ++scales)
        for (int per = scales_per_group(scales); per <= 4; ++per) {
            std::set<int> cols;
            for (int f = 0; f < 6 * scales; ++f) {
                const int col = layer0_kernel_column(f, scales, per);
                CHECK(col >= 0 && col < kEncIn && cols.insert(col).second);
                const int t = col / 16, g = (col % 16) / 4, r = col % 4;
                CHECK(layer0_source_feature(t, g, r, scales, per) == f);
            }
            int real = 0;
            for (int t = 0; t < 6; ++t)
                for (int g = 0; g < 4; ++g)
                    for (int r = 0; r < 4; ++r) real += layer0_source_feature(t, g, r, scales, per) >= 0;
            CHECK(real == 6 * scales);
        }
    for (int t = 0; t < 6; ++t)                                             // the full-width form is per = 4 at 16 scales
        for (int g = 0; g < 4; ++g)
            for (int r = 0; r < 4; ++r) CHECK(layer0_source_feature(t, g, r) == layer0_source_feature(t, g, r, 16, 4));
    // ---- dense layer 0 of the 4-tile kernels ----
    CHECK(layer0_dense(4, 1) && layer0_dense(4, 2) && !layer0_dense(4, 3) && !layer0_dense(8, 2) && !layer0_dense(16, 1));
    for (int scales = 1; scales <= 8; ++scales) {
        const int per = scales_per_group(scales);
        std::set<int> used;                                                  // old slots q = 4 t + r a lane group fills
        for (int g = 0; g < 4; ++g)
            for (int t = 0; t < 6; ++t)
                for (int r = 0; r < 4; ++r)
                    if (layer0_source_feature(t, g, r, scales, per) >= 0) used.insert(4 * t + r);
        std::set<int> dense;
        int last = -1;
        for (int d = 0; d < 16; ++d) {
            const int q = layer0_dense_source_slot(d);
            if (d >= 12) { CHECK(q == -1); continue; }
            CHECK(q > last);
            last = q;
            dense.insert(q);
        }
        for (int q : used) CHECK(dense.count(q) == 1);                       // nothing a network of <= 8 scales fills is lost
    }
    std::printf("layout ok\n");
    return 0;
}
"""


def test_layout_maps_are_consistent(tmp_path):
    src = tmp_path / "layout_check.cpp"
    src.write_text(PROGRAM)
    exe = tmp_path / "layout_check"
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "nerf_amd", "csrc"), str(src), "-o", str(exe)],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True)
    assert run.returncode == 0 and "layout ok" in run.stdout, run.stdout + run.stderr
