/* bgt.h -- drop-in name of the reader API header (reference bgt.h); see bgt_reader.h */
#include "bgt_reader.h"
