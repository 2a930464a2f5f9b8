/* stand-in for <mjxmacro.h>: nothing from it is used by the headers the binding parses */
