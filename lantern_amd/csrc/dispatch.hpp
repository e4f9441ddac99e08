// dispatch.hpp -- host-side dispatch of a kernel template on (metric code, lanes per row).
#pragma once
#include "device_common.hpp"

// dispatch on (metric, lanes per row)
#define LGPU_DISPATCH(metric, chunks, CALL)                                   \
    do {                                                                      \
        const int G_ = group_lanes_for(chunks);                               \
        switch(metric) {                                                      \
            case M_L2SQ:                                                      \
                switch(G_) { case 64: CALL(M_L2SQ, 64); break; case 32: CALL(M_L2SQ, 32); break; \
                             case 16: CALL(M_L2SQ, 16); break; default: CALL(M_L2SQ, 8); }       \
                break;                                                        \
            case M_COS:                                                       \
                switch(G_) { case 64: CALL(M_COS, 64); break; case 32: CALL(M_COS, 32); break;   \
                             case 16: CALL(M_COS, 16); break; default: CALL(M_COS, 8); }         \
                break;                                                        \
            case M_HAMMING:                                                   \
                switch(G_) { case 64: CALL(M_HAMMING, 64); break; case 32: CALL(M_HAMMING, 32); break; \
                             case 16: CALL(M_HAMMING, 16); break; default: CALL(M_HAMMING, 8); } \
                break;                                                        \
            case M_COS_B1:                                                    \
                switch(G_) { case 64: CALL(M_COS_B1, 64); break; case 32: CALL(M_COS_B1, 32); break; \
                             case 16: CALL(M_COS_B1, 16); break; default: CALL(M_COS_B1, 8); } \
                break;                                                        \
            case M_L2SQ_F16:                                                  \
                switch(G_) { case 64: CALL(M_L2SQ_F16, 64); break; case 32: CALL(M_L2SQ_F16, 32); break; \
                             case 16: CALL(M_L2SQ_F16, 16); break; default: CALL(M_L2SQ_F16, 8); } \
                break;                                                        \
            case M_COS_F16:                                                   \
                switch(G_) { case 64: CALL(M_COS_F16, 64); break; case 32: CALL(M_COS_F16, 32); break; \
                             case 16: CALL(M_COS_F16, 16); break; default: CALL(M_COS_F16, 8); } \
                break;                                                        \
            case M_L2SQ_I8:                                                   \
                switch(G_) { case 64: CALL(M_L2SQ_I8, 64); break; case 32: CALL(M_L2SQ_I8, 32); break; \
                             case 16: CALL(M_L2SQ_I8, 16); break; default: CALL(M_L2SQ_I8, 8); } \
                break;                                                        \
            case M_COS_I8:                                                    \
                switch(G_) { case 64: CALL(M_COS_I8, 64); break; case 32: CALL(M_COS_I8, 32); break; \
                             case 16: CALL(M_COS_I8, 16); break; default: CALL(M_COS_I8, 8); } \
                break;                                                        \
            default: return hipErrorInvalidValue;                             \
        }                                                                     \
    } while(0)


