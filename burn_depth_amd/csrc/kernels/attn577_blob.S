/* the gfx950 code object of kernels/attn577_gfx950.s as read-only bytes of the host library (assembled from the build directory) */
    .section .rodata
    .balign 4096
    .globl md_attn577_co
    .globl md_attn577_co_end
md_attn577_co:
    .incbin "attn577.co"
md_attn577_co_end:
    .section .note.GNU-stack,"",@progbits
