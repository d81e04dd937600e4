// Host-only: rsem_amd/csrc/host/model_host.hpp reads a .model file written by the reference and writes it again --
// tests/test_capi_cpu.py wants the same bytes back (the values are printed with the reference's %.10g / %.15g, which a
// parse-and-print leaves unchanged), i.e. the writer's layout is the reference's, character for character.
#include "../rsem_amd/csrc/host/model_host.hpp"
int main(int argc, char** argv) {
    if (argc < 3) return 2;
    rsemh::Model m;
    m.read(argv[1], 0);
    m.write(argv[2]);
    return 0;
}
