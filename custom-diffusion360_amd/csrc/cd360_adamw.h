// Tensor slots of one cd360_adamw_bf16 launch (the pointer table travels by value in the kernel arguments: 48 bytes per tensor, 4 KB limit).
// Mirrors CD360_ADAMW_MAX_TENSORS of include/cd360_hip.h.
#pragma once
#define CD360_ADAMW_MAX_TENSORS 64
