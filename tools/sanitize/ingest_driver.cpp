#include "badger_hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
// usage: ingest_driver <file> <reader threads> [segment bytes]
int main(int argc, char** argv)
{
    for (int rep = 0; rep < 3; ++rep) {
        bdg_ingest* g = nullptr;
        bdg_ingest_opts o;
        memset(&o, 0, sizeof(o));
        o.chunk_reads = 256; o.ring_chunks = 3; o.pinned = 0; o.threads = (uint32_t)atoi(argv[2]);
        o.segment_bytes = argc > 3 ? strtoull(argv[3], nullptr, 10) : 0;
        int rc = bdg_ingest_open_ex(argv[1], &o, &g);
        if (rc) { printf("open rc %d\n", rc); return 1; }
        unsigned long long n = 0, bytes = 0; int k = 0;
        for (;;) {
            bdg_ingest_chunk ch;
            rc = bdg_ingest_next(g, &ch);
            if (rc) { printf("next rc %d: %s\n", rc, bdg_ingest_error(g)); break; }
            if (ch.n == 0) break;
            n += ch.n; bytes += ch.total_bytes;
            bdg_ingest_release(g, ch.id);
            if (rep == 2 && ++k == 2) break;          // close in mid-file: threads must wind down
        }
        bdg_ingest_close(g);
        printf("reads %llu bytes %llu\n", n, bytes);
    }
    return 0;
}
