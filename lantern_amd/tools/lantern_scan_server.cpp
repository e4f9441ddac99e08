// lantern-scan-server -- standalone front end of the scan-side service in liblantern_gpu.so: loads a usearch-format
// index file (what `usearch_save` / the external indexing server produce, lantern_hnsw/src/hnsw/build.c:583) into HBM and
// serves the k-NN queries of many PostgreSQL backends in coalesced launches (lantern_amd/csrc/scan_server.cpp).
//   --index FILE --metric l2sq|cos|hamming --dim D --m M [--ef 64] [--ef-construction 128] [--quant-bits 32|16|8]
//   [--host 127.0.0.1] [--port 8997] [--max-batch 256] [--max-wait-us 150]
// `--dim` is the number of f32 scalars, or of BITS for hamming -- the reloption `dim` as Lantern passes it (scan.c:84-88).
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/lantern_gpu.h"

int main(int argc, char **argv)
{
    // one hardware queue per dispatcher lane (the HIP runtime's default of four makes lanes queue behind one another: index.cpp);
    // read when the runtime initialises, so it is set here, before the first call into the library; an explicit setting wins
    (void)::setenv("GPU_MAX_HW_QUEUES", "16", 0);
    std::string host = "127.0.0.1", index_path, metric = "l2sq";
    int         port = 8997, quant_bits = 32;
    size_t      dim = 0, m = 16, ef = 64, efc = 128, max_batch = 256;
    unsigned    max_wait_us = 150;
    for(int i = 1; i < argc; ++i) {
        auto val = [&](const char *name) -> const char * {
            if(std::strcmp(argv[ i ], name) == 0 && i + 1 < argc) return argv[ ++i ];
            return nullptr;
        };
        if(const char *v = val("--host")) host = v;
        else if(const char *v = val("--port")) port = std::atoi(v);
        else if(const char *v = val("--index")) index_path = v;
        else if(const char *v = val("--metric")) metric = v;
        else if(const char *v = val("--dim")) dim = (size_t)std::atoll(v);
        else if(const char *v = val("--m")) m = (size_t)std::atoll(v);
        else if(const char *v = val("--ef")) ef = (size_t)std::atoll(v);
        else if(const char *v = val("--ef-construction")) efc = (size_t)std::atoll(v);
        else if(const char *v = val("--quant-bits")) quant_bits = std::atoi(v);
        else if(const char *v = val("--max-batch")) max_batch = (size_t)std::atoll(v);
        else if(const char *v = val("--max-wait-us")) max_wait_us = (unsigned)std::atoi(v);
        else {
            std::fprintf(stderr,
                         "usage: %s --index FILE --metric l2sq|cos|hamming --dim D --m M [--ef 64] [--ef-construction 128]\n"
                         "          [--quant-bits 32|16|8] [--host H] [--port P] [--max-batch N] [--max-wait-us U]\n",
                         argv[ 0 ]);
            return 2;
        }
    }
    if(index_path.empty() || dim == 0) {
        std::fprintf(stderr, "--index and --dim are required\n");
        return 2;
    }
    usearch_init_options_t o;
    std::memset(&o, 0, sizeof(o));
    o.metric_kind = metric == "cos" ? usearch_metric_cos_k : metric == "hamming" ? usearch_metric_hamming_k : usearch_metric_l2sq_k;
    if(metric != "cos" && metric != "hamming" && metric != "l2sq") {
        std::fprintf(stderr, "unknown metric %s\n", metric.c_str());
        return 2;
    }
    o.quantization = o.metric_kind == usearch_metric_hamming_k ? usearch_scalar_b1_k  // options.c:137-158
                     : quant_bits == 16                        ? usearch_scalar_f16_k
                     : quant_bits == 8                         ? usearch_scalar_i8_k
                                                               : usearch_scalar_f32_k;
    o.dimensions = dim;
    o.connectivity = m;
    o.expansion_add = efc;
    o.expansion_search = ef;
    o.num_threads = 1;
    usearch_error_t err = nullptr;
    usearch_index_t index = usearch_init(&o, nullptr, &err);
    if(err) {
        std::fprintf(stderr, "%s\n", err);
        return 1;
    }
    usearch_load(index, index_path.c_str(), &err);
    if(err) {
        std::fprintf(stderr, "%s\n", err);
        return 1;
    }
    const size_t            n = usearch_size(index, &err);
    lantern_scan_server_t *s = lantern_scan_server_start(index, host.c_str(), port, max_batch, max_wait_us, &err);
    if(!s) {
        std::fprintf(stderr, "%s\n", err ? err : "cannot start the server");
        return 1;
    }
    std::printf("Scan server started on %s:%d: %zu vectors resident (%s)\n", host.c_str(), lantern_scan_server_port(s), n, lantern_gpu_version());
    std::fflush(stdout);
    for(;;) pause();
}
