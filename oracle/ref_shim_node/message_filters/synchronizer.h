#include <message_filters/subscriber.h>
