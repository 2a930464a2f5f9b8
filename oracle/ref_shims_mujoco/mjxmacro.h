/* shim: the reference includes <mjxmacro.h> but uses none of its macros */
