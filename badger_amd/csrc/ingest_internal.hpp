// Shared between ingest.cpp (the readers) and stage1.cpp (the native stage-1 driver): a chunk of parsed reads.
#pragma once

#include <cstddef>
#include <cstdint>

struct IngestChunk {
    uint8_t*  bases = nullptr;  size_t bases_cap = 0;      // concatenated ASCII bases (pinned when the reader was opened so)
    uint64_t* off = nullptr;    size_t off_cap = 0;        // n + 1 offsets into bases
    char*     ids = nullptr;    size_t ids_cap = 0;        // concatenated read ids
    uint64_t* id_off = nullptr; size_t id_off_cap = 0;     // n + 1 offsets into ids
    uint32_t  n = 0;
    uint64_t  bases_bytes = 0, ids_bytes = 0;
    uint32_t  views = 0;                                   // views of this chunk that the consumer has not released yet
};
