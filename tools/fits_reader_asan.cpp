// Developer harness: the native FITS readers (lfd_amd/csrc/fits_reader.h, host code only) under AddressSanitizer + UBSan on
// the files given on the command line -- truncated and hostile headers must come back as a status, never as a read outside
// the file buffer.  Built and run by tools/fits_reader_sanitize.sh (GPU sanitizers are not available on this pool).
#include <stdio.h>
#include <vector>
#include "../lfd_amd/csrc/fits_reader.h"

int main(int argc, char **argv) {
    const int max_obj = 64, h = 64, w = 96;
    for (int i = 1; i < argc; i++) {
        const char *p = argv[i];
        std::vector<float> f5[4];
        for (auto &v : f5) v.resize((size_t)max_obj * 5);
        std::vector<int32_t> no(max_obj), nd(max_obj);
        int32_t count = 0, st = 0, fst = 0, hl = 0;
        int rc = lfdmi_fits_read_photoobj(&p, 1, max_obj, f5[0].data(), f5[1].data(), f5[2].data(), f5[3].data(), no.data(), nd.data(), &count, 1, &st);
        std::vector<char> frame((size_t)h * w * 4), hdr(8640);
        int rc2 = lfdmi_fits_read_frames(&p, 1, h, w, frame.data(), 1, &fst, hdr.data(), (int)hdr.size(), &hl);
        printf("%-40s photoobj rc %d status %d rows %d | frame rc %d status %d\n", p, rc, st, count, rc2, fst);
    }
    return 0;
}
