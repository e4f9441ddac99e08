// lantern-index-server -- standalone front end of the external indexing server in liblantern_gpu.so.
// Same flags as `lantern-cli start-indexing-server` (lantern_cli/src/external_index/cli.rs:126-151):
//   --host 0.0.0.0 --port 8998 --status-port 8999 --tmp-dir /tmp [--cert cert.pem --key key.pem]
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/lantern_gpu.h"

int main(int argc, char **argv)
{
    std::string host = "0.0.0.0", tmp = "/tmp", cert, key;
    int         port = 8998, status_port = 8999;
    for(int i = 1; i < argc; ++i) {
        auto val = [&](const char *name) -> const char * {
            if(std::strcmp(argv[ i ], name) == 0 && i + 1 < argc) return argv[ ++i ];
            return nullptr;
        };
        if(const char *v = val("--host")) host = v;
        else if(const char *v = val("--port")) port = std::atoi(v);
        else if(const char *v = val("--status-port")) status_port = std::atoi(v);
        else if(const char *v = val("--tmp-dir")) tmp = v;
        else if(const char *v = val("--cert")) cert = v;
        else if(const char *v = val("--key")) key = v;
        else {
            std::fprintf(stderr, "usage: %s [--host H] [--port P] [--status-port P] [--tmp-dir D] [--cert cert.pem --key key.pem]\n", argv[ 0 ]);
            return 2;
        }
    }
    if(lantern_gpu_device_count() <= 0) std::fprintf(stderr, "warning: no HIP device visible; every build request will fail\n");
    usearch_error_t         err = nullptr;
    lantern_index_server_t *s = lantern_index_server_start_tls(host.c_str(), port, status_port, tmp.c_str(), cert.empty() ? nullptr : cert.c_str(),
                                                               key.empty() ? nullptr : key.c_str(), &err);
    if(!s) {
        std::fprintf(stderr, "%s\n", err ? err : "cannot start the server");
        return 1;
    }
    std::printf("External Indexing Server started on %s:%d (%s)\n", host.c_str(), lantern_index_server_port(s), lantern_gpu_version());
    std::fflush(stdout);
    for(;;) pause();
}
