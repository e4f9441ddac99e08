// lantern-index-server -- standalone front end of the external indexing server in liblantern_gpu.so.
// Same flags as `lantern-cli start-indexing-server` (lantern_cli/src/external_index/cli.rs:126-151):
//   --host 0.0.0.0 --port 8998 --status-port 8999 --tmp-dir /tmp     (--cert/--key: TLS is not offered)
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/lantern_gpu.h"

int main(int argc, char **argv)
{
    std::string host = "0.0.0.0", tmp = "/tmp";
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
        else if(std::strcmp(argv[ i ], "--cert") == 0 || std::strcmp(argv[ i ], "--key") == 0) {
            std::fprintf(stderr, "TLS is not supported by this server\n");
            return 2;
        } else {
            std::fprintf(stderr, "usage: %s [--host H] [--port P] [--status-port P] [--tmp-dir D]\n", argv[ 0 ]);
            return 2;
        }
    }
    if(lantern_gpu_device_count() <= 0) std::fprintf(stderr, "warning: no HIP device visible; every build request will fail\n");
    usearch_error_t         err = nullptr;
    lantern_index_server_t *s = lantern_index_server_start(host.c_str(), port, status_port, tmp.c_str(), &err);
    if(!s) {
        std::fprintf(stderr, "%s\n", err ? err : "cannot start the server");
        return 1;
    }
    std::printf("External Indexing Server started on %s:%d (%s)\n", host.c_str(), lantern_index_server_port(s), lantern_gpu_version());
    std::fflush(stdout);
    for(;;) pause();
}
